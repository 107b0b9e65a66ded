#!/bin/bash
# round 5, session 9: trace + histogram of a large component in one launch; bulk witness copy
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_prove.py tests/test_gpu_workloads.py -x -q -m gpu -k "fibonacci_proof or configs1 or u32 or felt or workload or device_tail" > gpurun_out/r05j_first.txt 2>&1 || { tail -15 gpurun_out/r05j_first.txt; echo "first tests failed: stopping"; exit 1; }
tail -2 gpurun_out/r05j_first.txt
CM_HOST_MARKS=1 python tools/lone_loop.py 2>&1 | grep "tail finish\|trace_gen" | tail -4
for r in 1 2 3 4; do
  for v in "CM_TRACE_HIST_FUSE=0" "CM_TRACE_HIST_FUSE=1"; do
    echo "$v $(env $v timeout 120 python tools/lone_loop.py 2>&1 | tail -1)"; done
done > gpurun_out/r05j_ab_trace_hist_fuse.txt
cat gpurun_out/r05j_ab_trace_hist_fuse.txt
GAPS_HEAD=8 tools/gaps.sh r05j --list | head -8
awk '$1+0<520' gpurun_out/r05j_gaps.txt | grep " dur " | head -50
