#!/bin/bash
# pipelined throughput (cm_prove_many, 4 in flight) of several library builds inside ONE GPU session
R=$1; shift
for i in $(seq $R); do
  for L in "$@"; do
    echo -n "$(basename $L) "
    CAIROM_HIP_LIB=$L python bench.py --steps 4 --warmup 2 --no-cpu-baseline --pipelined 4 --no-end-to-end --no-kprof 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['pipelined']['ms_per_proof'],3))"
  done
done
