#!/bin/bash
# ms_per_step and every instrumented kernel class of the default workload, R runs:  tools/classes.sh [R] [filter-regex]
R=${1:-2}; F=${2:-.}
for i in $(seq $R); do
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline --pipelined 0 --no-end-to-end --alt-fib-n 0 2>/dev/null | F="$F" python -c "
import sys,json,os,re
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print(round(d['ms_per_step'],3), 'idle', d.get('gpu_idle_ms'), {n:round(v['ms_per_step'],3) for n,v in k.items() if re.search(os.environ['F'],n)})"
done
