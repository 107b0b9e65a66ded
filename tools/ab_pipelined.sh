#!/bin/bash
# like tools/ab_vals.sh, for the throughput mode: tools/ab_pipelined.sh VAR "v1 v2" [rounds]  -> lone ms, ms per proof with 4 in flight (32 proofs)
VAR=$1; VALS=$2; R=${3:-3}
for i in $(seq $R); do
  for v in $VALS; do
    if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
    echo -n "$VAR=$v "
    python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-end-to-end --alt-fib-n 0 --big-fib-n 0 --no-kprof 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['pipelined']['ms_per_proof'],3))"
  done
done
