#!/bin/bash
# phase_ms of several library builds inside ONE GPU session: tools/abphase.sh rounds lib1.so lib2.so ...
R=$1; shift
for i in $(seq $R); do
  for L in "$@"; do
    echo -n "$(basename $L) "
    CAIROM_HIP_LIB=$L python bench.py --steps 10 --warmup 2 --no-cpu-baseline --pipelined 0 --no-end-to-end --no-kprof 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['phase_ms'].items()})"
  done
done
