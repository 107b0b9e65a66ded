#!/bin/bash
# A/B of two library builds inside ONE GPU session, headline only: tools/ab2.sh A.so B.so [rounds]
A=$1; B=$2; R=${3:-4}
for i in $(seq $R); do
  for L in $A $B; do
    echo -n "$(basename $L) "
    CAIROM_HIP_LIB=$L python bench.py --steps 10 --warmup 2 --no-cpu-baseline --pipelined 0 --no-end-to-end --no-kprof 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3))"
  done
done
