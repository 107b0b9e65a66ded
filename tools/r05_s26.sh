#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-end-to-end --alt-fib-n 0 --big-fib-n 0 --pipelined 0"
for r in 1 2 3; do
  for v in 1 0; do
    CM_KPROF_EXT=$v $B > gpurun_out/r06f_kext$v.json 2> gpurun_out/r06f_kext$v.err
    python - <<PY
import json
d=json.load(open("gpurun_out/r06f_kext$v.json"))
r=d["roofline"]
print("CM_KPROF_EXT=$v ms_per_step %.3f  class %s avg_us %.2f launches %s frac %.4f alu %.3f" % (d["ms_per_step"], r["kernel"], r.get("avg_launch_us", 0), r.get("launches_timed"), r["frac"], r["alu"]["frac_of_register_only"]))
PY
  done
done
