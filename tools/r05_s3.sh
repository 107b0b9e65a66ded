#!/bin/bash
# round 5, session 3: tail with the looped grind + linear host mirror; the flag-join lab
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 120 tools/flagjoin_lab > gpurun_out/r05c_flagjoin_lab.txt 2>&1
cat gpurun_out/r05c_flagjoin_lab.txt
timeout 1200 python -m pytest tests/test_gpu_prove.py -x -q -m gpu -k "device_tail or fibonacci_proof or configs1 or metric_config" > gpurun_out/r05c_tests.txt 2>&1
tail -3 gpurun_out/r05c_tests.txt
for r in 1 2 3; do
  for v in "CM_DEVICE_TAIL=0 CM_OODS_SPLIT=1000" "CM_DEVICE_TAIL=1 CM_OODS_SPLIT=780"; do
    echo "$v $(env $v python tools/lone_loop.py 2>&1 | tail -1)"; done
done > gpurun_out/r05c_ab.txt
cat gpurun_out/r05c_ab.txt
CM_HOST_MARKS=1 python tools/lone_loop.py 2> gpurun_out/r05c_host_marks.txt > /dev/null
tail -12 gpurun_out/r05c_host_marks.txt | cut -c1-160
GAPS_HEAD=12 tools/gaps.sh r05c --list | head -12
grep -n "k_fri_tail" -A6 gpurun_out/r05c_gaps.txt | tail -8
