#!/bin/bash
# every kept switch of the round, alternating blocks inside one process
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python tools/ab_switch.py --reps 16 defer_teardown flag_join flag_fork commit_prep_early trace_hist_fuse logup_defer oods_poll oods_host_write \
   stage_copy_kernel stage_lazy_events oods_split=780,1000 oods_split=700,780 oods_split=850,780 fri_top_fuse cpu_affinity_sticky > gpurun_out/r06a_ab_switch.txt 2>&1
cat gpurun_out/r06a_ab_switch.txt
