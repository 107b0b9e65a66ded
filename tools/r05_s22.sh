#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python tools/ab_switch.py --reps 16 cpu_affinity_sticky defer_teardown cpu_affinity_sticky > gpurun_out/r05z2_ab_switch.txt 2>&1
cat gpurun_out/r05z2_ab_switch.txt
CM_CPU_AFFINITY=2 CM_HOST_MARKS=1 python tools/lone_loop.py 2>&1 | grep "setup\|~\|destroyed\|between\|ms per\|finish\|constructed" | tail -12
