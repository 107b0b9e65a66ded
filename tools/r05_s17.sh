#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python tools/ab_switch.py --reps 12 oods_host_write oods_poll stage_copy_kernel stage_lazy_events oods_poll > gpurun_out/r05w_ab_switch.txt 2>&1
cat gpurun_out/r05w_ab_switch.txt
tools/gaps.sh r05w --list > /dev/null; awk '$2=="us" && $1>6300 && $1<7400' gpurun_out/r05w_gaps.txt | head -40
