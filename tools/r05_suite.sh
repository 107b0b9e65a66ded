#!/bin/bash
# the whole GPU suite + default bench on one box: tools/r05_suite.sh <tag>
tag=${1:-r05x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 3000 python -m pytest tests/ -x -q -m gpu > gpurun_out/${tag}_suite.txt 2>&1
tail -4 gpurun_out/${tag}_suite.txt
python bench.py > gpurun_out/${tag}_bench_n1.json 2> gpurun_out/${tag}_bench_n1.err
python - <<PY
import json
d=json.load(open("gpurun_out/${tag}_bench_n1.json"))
print("ms_per_step", d["ms_per_step"], "value", d["value"], "pipelined", (d.get("pipelined") or {}).get("ms_per_proof"))
print({k: round(v,3) for k,v in d["phase_ms"].items()})
PY
