#!/bin/bash
# round 5, session 8: AVX-512 host Blake2s (mix_felts of the sampled values), tree-2 prep early; timeline
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
grep -m1 "model name" /proc/cpuinfo; grep -c avx512vl /proc/cpuinfo
timeout 300 python -m pytest tests/test_gpu_prove.py -x -q -m gpu -k "fibonacci_proof or configs1" > gpurun_out/r05i_first.txt 2>&1 || { tail -15 gpurun_out/r05i_first.txt; echo "first test failed: stopping"; exit 1; }
for v in "CM_HOST_B2S_NO_AVX512=1" "CM_X=1"; do
  echo "== $v"; env $v CM_HOST_MARKS=1 python tools/lone_loop.py 2>&1 | grep "oods:\|tail: host replay\|tail finish" | tail -8
done
for r in 1 2 3; do
  for v in "CM_HOST_B2S_NO_AVX512=1" "CM_X=1"; do
    echo "$v $(env $v timeout 120 python tools/lone_loop.py 2>&1 | tail -1)"; done
done > gpurun_out/r05i_ab_avx512.txt
cat gpurun_out/r05i_ab_avx512.txt
GAPS_HEAD=8 tools/gaps.sh r05i --list | head -8
