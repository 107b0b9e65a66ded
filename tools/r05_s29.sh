#!/bin/bash
# where the CPU oracle stops scaling on the GPU box's host (phase times at 16 / 32 / 64 threads)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for th in 32 64 128; do
  echo "== OMP_NUM_THREADS=$th"
  ORC_TIMING=1 OMP_NUM_THREADS=$th OMP_PROC_BIND=close OMP_PLACES=cores python oracle/cpu_baseline.py --fib-n 419000 --reps 2 2>&1 | tail -14
done > gpurun_out/r06l_oracle_scaling.txt 2>&1
cat gpurun_out/r06l_oracle_scaling.txt
