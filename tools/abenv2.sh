#!/bin/bash
# headline under several environment settings inside ONE GPU session, interleaved: tools/abenv2.sh rounds "VAR=1" "VAR2=1" "" ...
R=$1; shift
for i in $(seq $R); do
  for E in "$@"; do
    echo -n "[$E] "
    env $E python bench.py --steps 10 --warmup 2 --no-cpu-baseline --pipelined 0 --no-end-to-end --no-kprof 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['phase_ms'].items() if 'commit' in k or k=='constraints' or k=='interaction_gen'})"
  done
done
