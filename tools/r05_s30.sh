#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_prove.py -x -q -m gpu > gpurun_out/r06m_tests.txt 2>&1; tail -2 gpurun_out/r06m_tests.txt
grep -q failed gpurun_out/r06m_tests.txt && exit 1
CM_CPU_AFFINITY=2 CM_HOST_MARKS=1 python tools/lone_loop.py 2>&1 | grep "between\|destroyed\|ms per\|oods: overlapped" | tail -8
for r in 1 2 3; do echo "$(CM_CPU_AFFINITY=2 timeout 120 python tools/lone_loop.py 2>&1 | tail -1)"; done
