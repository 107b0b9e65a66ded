#!/bin/bash
# round 5, session 2: device tail (rank sort, wave scans) + chunked OODS values with streamed mix_felts
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_prove.py tests/test_gpu_framing.py -x -q -m gpu -k "device_tail or fibonacci_proof or configs1 or metric_config or log_blowup or framing" > gpurun_out/r05b_tests.txt 2>&1
tail -5 gpurun_out/r05b_tests.txt
for r in 1 2 3; do
  for v in "CM_DEVICE_TAIL=0 CM_OODS_SPLIT=1000" "CM_DEVICE_TAIL=1 CM_OODS_SPLIT=1000" "CM_DEVICE_TAIL=1 CM_OODS_SPLIT=780" "CM_DEVICE_TAIL=1 CM_OODS_SPLIT=650" "CM_DEVICE_TAIL=1 CM_OODS_SPLIT=850"; do
    echo "$v $(env $v python tools/lone_loop.py 2>&1 | tail -1)"; done
done > gpurun_out/r05b_ab.txt
cat gpurun_out/r05b_ab.txt
CM_HOST_MARKS=1 python tools/lone_loop.py 2> gpurun_out/r05b_host_marks.txt > /dev/null
tail -45 gpurun_out/r05b_host_marks.txt | cut -c1-160 > gpurun_out/r05b_host_marks_tail.txt
GAPS_HEAD=30 tools/gaps.sh r05b --list
