#!/bin/bash
# A/B of an environment switch inside ONE GPU session: tools/ab_env.sh VAR [rounds] [extra bench args]
# prints ms_per_step of `bench.py` with VAR unset (A) and VAR=1 (B), alternating.
VAR=$1; R=${2:-4}; shift; shift
for i in $(seq $R); do
  for m in A B; do
    if [ $m = B ]; then export $VAR=1; else unset $VAR; fi
    echo -n "$m "
    python bench.py --steps 10 --warmup 2 --no-cpu-baseline --pipelined 0 --no-end-to-end --no-kprof "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3))"
  done
done
