#!/bin/bash
timeout 120 python tools/jobs/warm_probe.py 2>&1 < /dev/null | tail -3
timeout 120 python tools/jobs/warm_probe.py 2>&1 < /dev/null | tail -2
