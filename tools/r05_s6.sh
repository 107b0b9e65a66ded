#!/bin/bash
# round 5, session 6: flag fork (side streams start behind a flag of the main stream instead of an event)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_prove.py -x -q -m gpu -k "fibonacci_proof" > gpurun_out/r05g_first.txt 2>&1 || { tail -15 gpurun_out/r05g_first.txt; echo "first test failed: stopping"; exit 1; }
for r in 1 2 3 4; do
  for v in "CM_FLAG_FORK=0" "CM_FLAG_FORK=1"; do
    echo "$v $(env $v timeout 120 python tools/lone_loop.py 2>&1 | tail -1)"; done
done > gpurun_out/r05g_ab_flag_fork.txt
cat gpurun_out/r05g_ab_flag_fork.txt
timeout 1500 python -m pytest tests/test_gpu_prove.py tests/test_gpu_components.py tests/test_gpu_sharded.py tests/test_gpu_adapter.py -x -q -m gpu -k "not 2pow24 and not at_scale" > gpurun_out/r05g_tests.txt 2>&1
tail -3 gpurun_out/r05g_tests.txt
GAPS_HEAD=14 tools/gaps.sh r05g --list | head -14
