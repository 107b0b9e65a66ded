#!/bin/bash
# kernel-class times (kprof) of several library builds inside ONE GPU session: tools/abclass.sh rounds "<class regex>" lib1.so lib2.so ...
R=$1; PAT=$2; shift; shift
for i in $(seq $R); do
  for L in "$@"; do
    echo -n "$(basename $L) "
    export PAT
    CAIROM_HIP_LIB=$L python bench.py --steps 8 --warmup 2 --no-cpu-baseline --pipelined 0 --no-end-to-end 2>/dev/null | python -c "
import sys,json,os,re
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print(round(d['ms_per_step'],3), {n: round(v['ms_per_step'],3) for n,v in k.items() if re.search(os.environ['PAT'], n)})"
  done
done
