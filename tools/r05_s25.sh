#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_prove.py tests/test_gpu_workloads.py -x -q -m gpu > gpurun_out/r06e_tests.txt 2>&1; tail -3 gpurun_out/r06e_tests.txt
grep -q failed gpurun_out/r06e_tests.txt && exit 1
timeout 900 python tools/ab_switch.py --reps 16 defer_teardown > gpurun_out/r06e_ab_switch.txt 2>&1
cat gpurun_out/r06e_ab_switch.txt
CM_HOST_MARKS=1 python tools/lone_loop.py 2>&1 | grep "teardown\|~\|destroyed\|between\|ms per\|finish\|copied" | tail -12
