#!/bin/bash
timeout 200 python tools/jobs/seed_probe_f32.py 500388 2>&1 < /dev/null | tail -6
