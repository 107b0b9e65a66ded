#!/bin/bash
# OODS polling + stage_upload as a kernel / lazy ring events: tests, A/B, timeline
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_prove.py tests/test_gpu_adapter.py -x -q -m gpu > gpurun_out/r05v_tests.txt 2>&1; tail -3 gpurun_out/r05v_tests.txt
grep -q passed gpurun_out/r05v_tests.txt || exit 1
{
for r in 1 2 3; do
  echo "all on                  $(timeout 120 python tools/lone_loop.py 2>&1 | tail -1)"
  echo "CM_OODS_POLL=0          $(CM_OODS_POLL=0 timeout 120 python tools/lone_loop.py 2>&1 | tail -1)"
  echo "CM_STAGE_COPY_KERNEL=0  $(CM_STAGE_COPY_KERNEL=0 timeout 120 python tools/lone_loop.py 2>&1 | tail -1)"
  echo "CM_STAGE_LAZY_EVENTS=0  $(CM_STAGE_LAZY_EVENTS=0 timeout 120 python tools/lone_loop.py 2>&1 | tail -1)"
  echo "all three off           $(CM_OODS_POLL=0 CM_STAGE_COPY_KERNEL=0 CM_STAGE_LAZY_EVENTS=0 timeout 120 python tools/lone_loop.py 2>&1 | tail -1)"
done
} > gpurun_out/r05v_ab.txt 2>&1
cat gpurun_out/r05v_ab.txt
GAPS_HEAD=8 tools/gaps.sh r05v
CM_OODS_POLL=0 CM_STAGE_COPY_KERNEL=0 CM_STAGE_LAZY_EVENTS=0 GAPS_HEAD=3 tools/gaps.sh r05v_off
