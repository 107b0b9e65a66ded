#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python tools/ab_switch.py --reps 12 merkle_npw=-1,2 merkle_npw=-1,4 merkle_npw=-1,8 logup_width=4,3 logup_width=4,5 fork_width=0,5 fork_width=0,6 > gpurun_out/r06k_ab_switch.txt 2>&1
cat gpurun_out/r06k_ab_switch.txt
