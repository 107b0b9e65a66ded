#!/bin/bash
# policy choices of rounds 2-4, alternating blocks inside one process
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_prove.py -x -q -m gpu -k "fibonacci_proof or metric_config" > gpurun_out/r06h_tests.txt 2>&1; tail -2 gpurun_out/r06h_tests.txt
timeout 1500 python tools/ab_switch.py --reps 12 fork_main merkle_npw=-1,0 fork_width=0,4 pp_side tree0_prio=-1,0 tree1_first logup_width=4,6 logup_width=4,2 \
   quot_rows=2,1 quot_rows=2,4 fri_fold_leaf fft_fused commit_pipe fft_chunk_mb=0,64 pace=-1,0 pace=-1,1 pace_early > gpurun_out/r06h_ab_switch.txt 2>&1
cat gpurun_out/r06h_ab_switch.txt
