#!/bin/bash
# round 5, session 10: the flag fork / flag join with four proofs in flight; the LogUp golden test on the GPU
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_logup_golden.py -x -q -m gpu > gpurun_out/r05l_logup.txt 2>&1; tail -4 gpurun_out/r05l_logup.txt
tools/ab_pipelined.sh CM_FLAG_FORK "0 1" 3 > gpurun_out/r05l_ab_pipelined_flag_fork.txt; cat gpurun_out/r05l_ab_pipelined_flag_fork.txt
tools/ab_pipelined.sh CM_FLAG_JOIN "0 1" 2 > gpurun_out/r05l_ab_pipelined_flag_join.txt; cat gpurun_out/r05l_ab_pipelined_flag_join.txt
