#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_prove.py -x -q -m gpu -k "device_tail or fibonacci_proof or configs1 or metric_config" > gpurun_out/r05r_tests.txt 2>&1; tail -3 gpurun_out/r05r_tests.txt
CM_HOST_MARKS=1 python tools/lone_loop.py 2>&1 | grep "tail\|decommit" | tail -7
for r in 1 2 3; do echo "$(timeout 120 python tools/lone_loop.py 2>&1 | tail -1)"; done
GAPS_HEAD=4 tools/gaps.sh r05r --list | head -4
