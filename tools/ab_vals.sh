#!/bin/bash
# A/B of several VALUES of one environment variable inside ONE GPU session (box-to-box spread is ~4-8 %):
#   tools/ab_vals.sh VAR "v1 v2 v3" [rounds] [extra bench args]     ("-" = unset)
# prints ms_per_step, the phase times (PH=regex picks phases, default all) and the large kernel classes for every value, alternating.
VAR=$1; VALS=$2; R=${3:-3}; shift; shift; shift
for i in $(seq $R); do
  for v in $VALS; do
    if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
    echo -n "$VAR=$v "
    python bench.py --steps ${STEPS:-8} --warmup 2 --no-cpu-baseline --pipelined 0 --no-end-to-end --alt-fib-n 0 "$@" 2>/dev/null | PH="${PH:-.}" python -c "
import sys,json,os,re
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print(round(d['ms_per_step'],3), {n:round(v,3) for n,v in d['phase_ms'].items() if re.search(os.environ['PH'],n)}, {n:round(v['ms_per_step'],3) for n,v in k.items() if v['ms_per_step']>0.25})"
  done
done
