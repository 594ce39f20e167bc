#!/bin/bash
timeout 600 python tools/jobs/slow_seed_probe.py 900004 0x1.ecf8ae0000000p-3 2>&1 < /dev/null | tail -12
