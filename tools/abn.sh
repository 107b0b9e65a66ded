#!/bin/bash
# A/B of N library builds inside ONE GPU session: tools/abn.sh rounds lib1.so lib2.so ...   (PH=regex picks phases, KC=regex picks kernel classes)
R=$1; shift
for i in $(seq $R); do
  for L in "$@"; do
    echo -n "$(basename $L) "
    CAIROM_HIP_LIB=$L python bench.py --steps ${STEPS:-8} --warmup 2 --no-cpu-baseline --pipelined 0 --no-end-to-end --alt-fib-n 0 2>/dev/null | PH="${PH:-.}" KC="${KC:-.}" python -c "
import sys,json,os,re
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print(round(d['ms_per_step'],3), {n:round(v,3) for n,v in d['phase_ms'].items() if re.search(os.environ['PH'],n)}, {n:round(v['ms_per_step'],3) for n,v in k.items() if re.search(os.environ['KC'],n)})"
  done
done
