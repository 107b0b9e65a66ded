#!/bin/bash
# round 5, session 5: fold + transcript step inside the tree-top launch of the small FRI layers
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_prove.py -x -q -m gpu -k "fibonacci_proof" > gpurun_out/r05f_first.txt 2>&1 || { tail -15 gpurun_out/r05f_first.txt; echo "first test failed: stopping"; exit 1; }
timeout 1500 python -m pytest tests/test_gpu_prove.py tests/test_gpu_fri_quotients.py tests/test_gpu_sharded.py tests/test_gpu_framing.py -x -q -m gpu -k "not 2pow24 and not at_scale" > gpurun_out/r05f_tests.txt 2>&1
tail -3 gpurun_out/r05f_tests.txt
for r in 1 2 3 4; do
  for v in "CM_FRI_TOP_FUSE=0" "CM_FRI_TOP_FUSE=1"; do
    echo "$v $(env $v timeout 120 python tools/lone_loop.py 2>&1 | tail -1)"; done
done > gpurun_out/r05f_ab_fri_top_fuse.txt
cat gpurun_out/r05f_ab_fri_top_fuse.txt
GAPS_HEAD=6 tools/gaps.sh r05f --list | head -6
awk '$1+0>7700' gpurun_out/r05f_gaps.txt | grep " dur " | tail -60
