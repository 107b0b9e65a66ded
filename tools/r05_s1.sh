#!/bin/bash
# round 5, session 1: the device-side proof tail — parity, same-session A/B against the host-driven tail, traced idle time
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_prove.py -x -q -m gpu -k "device_tail or fibonacci_proof or configs1 or metric_config or log_blowup" > gpurun_out/r05a_tests.txt 2>&1
tail -5 gpurun_out/r05a_tests.txt
for r in 1 2 3 4; do
  for v in 0 1; do echo "CM_DEVICE_TAIL=$v $(CM_DEVICE_TAIL=$v python tools/lone_loop.py 2>&1 | tail -1)"; done
done > gpurun_out/r05a_ab_device_tail.txt
cat gpurun_out/r05a_ab_device_tail.txt
CM_HOST_TRACE=1 python tools/lone_loop.py 2> gpurun_out/r05a_host_trace.txt > /dev/null
tail -60 gpurun_out/r05a_host_trace.txt | cut -c1-160 > gpurun_out/r05a_host_trace_tail.txt
GAPS_HEAD=30 tools/gaps.sh r05a --list
