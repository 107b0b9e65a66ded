#!/bin/bash
# round 5, session 7: tree 2's host-side commit preparation under the LogUp kernels
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_prove.py -x -q -m gpu -k "fibonacci_proof or configs1" > gpurun_out/r05h_first.txt 2>&1 || { tail -15 gpurun_out/r05h_first.txt; echo "first test failed: stopping"; exit 1; }
for r in 1 2 3 4; do
  for v in "CM_COMMIT_PREP_EARLY=0" "CM_COMMIT_PREP_EARLY=1"; do
    echo "$v $(env $v timeout 120 python tools/lone_loop.py 2>&1 | tail -1)"; done
done > gpurun_out/r05h_ab_prep_early.txt
cat gpurun_out/r05h_ab_prep_early.txt
GAPS_HEAD=10 tools/gaps.sh r05h --list | head -10
