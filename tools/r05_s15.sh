#!/bin/bash
# OODS values: polling the landing words against event / stream synchronisation
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_prove.py -x -q -m gpu -k "fibonacci_proof or metric_config or device_tail or configs1" > gpurun_out/r05v_tests.txt 2>&1; tail -3 gpurun_out/r05v_tests.txt
grep -q passed gpurun_out/r05v_tests.txt || exit 1
{
for r in 1 2 3 4; do
  for v in 1 0; do echo "CM_OODS_POLL=$v $(CM_OODS_POLL=$v timeout 120 python tools/lone_loop.py 2>&1 | tail -1)"; done
done
echo "-- host marks, poll"; CM_HOST_MARKS=1 python tools/lone_loop.py 2>&1 | grep "oods:\|quotients:" | tail -9
echo "-- host marks, events"; CM_OODS_POLL=0 CM_HOST_MARKS=1 python tools/lone_loop.py 2>&1 | grep "oods:\|quotients:" | tail -9
} > gpurun_out/r05v_ab_oods_poll.txt 2>&1
cat gpurun_out/r05v_ab_oods_poll.txt
GAPS_HEAD=6 tools/gaps.sh r05v
