#!/bin/bash
# development tool: lone / four-in-flight ms per proof under GPU_MAX_HW_QUEUES = 4 (default), 8, 6, alternating processes in one session
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do
  for q in 4 8 6; do
    echo -n "GPU_MAX_HW_QUEUES=$q "
    GPU_MAX_HW_QUEUES=$q python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-end-to-end --alt-fib-n 0 --big-fib-n 0 --cached-setup-steps 0 --sharded-one-rank-blocks 0 --no-kprof --pipelined 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['pipelined']['ms_per_proof'],3))"
  done
done
