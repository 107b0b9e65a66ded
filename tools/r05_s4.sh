#!/bin/bash
# round 5, session 4: flag join (Fork::join as flag kernels + one collector) — parity and same-session A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 100 python -m pytest tests/test_gpu_prove.py -x -q -m gpu -k "fibonacci_proof" > gpurun_out/r05d_first.txt 2>&1 || { tail -5 gpurun_out/r05d_first.txt; echo "first test failed: stopping"; exit 1; }
timeout 1500 python -m pytest tests/test_gpu_prove.py tests/test_gpu_components.py tests/test_gpu_sharded.py -x -q -m gpu -k "not 2pow24 and not at_scale" > gpurun_out/r05d_tests.txt 2>&1
tail -3 gpurun_out/r05d_tests.txt
for r in 1 2 3 4; do
  for v in "CM_FLAG_JOIN=0" "CM_FLAG_JOIN=1"; do
    echo "$v $(env $v timeout 120 python tools/lone_loop.py 2>&1 | tail -1)"; done
done > gpurun_out/r05d_ab_flag_join.txt
cat gpurun_out/r05d_ab_flag_join.txt
GAPS_HEAD=14 tools/gaps.sh r05d --list | head -14
