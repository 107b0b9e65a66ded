#!/bin/bash
# A/B timing of two builds of the library inside ONE GPU session (box-to-box variance is ~4 %):
#   tools/ab.sh /path/A.so /path/B.so [rounds]
A=$1; B=$2; R=${3:-3}
for i in $(seq $R); do
  for L in $A $B; do
    CAIROM_HIP_LIB=$L python bench.py --steps 6 --warmup 2 --no-cpu-baseline --pipelined 0 --alt-fib-n 0 --no-end-to-end 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['roofline']['kernels'];print('$L'.split('/')[-1], round(d['ms_per_step'],3), {n:round(v['ms_per_step'],3) for n,v in k.items() if True})"
  done
done
