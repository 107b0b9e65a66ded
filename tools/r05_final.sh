#!/bin/bash
# round 5 measurement set of HEAD: tools/r05_final.sh <tag>
tag=${1:-r05p}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
tools/measure_round.sh $tag
python - <<PY
import json
d=json.load(open("gpurun_out/${tag}_bench_n1.json"))
print("ms_per_step", d["ms_per_step"], "value", d["value"], "pipelined", (d.get("pipelined") or {}).get("ms_per_proof"))
print({k: round(v,3) for k,v in d["phase_ms"].items()})
r=d["roofline"]; print(r["bound"], r["kernel"], r["frac"], r["alu"]["frac_of_register_only"] if r.get("alu") else None)
print(r["whole_path_model"], r.get("whole_path_model_larger_sizes"))
b=json.load(open("gpurun_out/${tag}_bench_big_legs.json"))
print(b["roofline"].get("whole_path_model_larger_sizes"))
PY
head -3 gpurun_out/${tag}_gaps.txt
