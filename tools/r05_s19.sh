#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python tools/ab_switch.py --reps 16 --phases oods_poll stage_copy_kernel > gpurun_out/r05y_ab_switch.txt 2>&1
cat gpurun_out/r05y_ab_switch.txt
