#!/bin/bash
# Quick GPU iteration loop: whole-proof bit-exactness vs the oracle (two small segments), then the bench.
# usage (on the GPU box): tools/gpu_quick.sh <tag> [extra bench args]
tag=${1:-x}; shift
python -m pytest tests/test_gpu_prove.py -m gpu -x -q -k "fibonacci_proof_bit_exact" 2>&1 | tail -3
python bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/bench_r1_$tag.json 2> gpurun_out/bench_r1_$tag.err
tail -c 300 gpurun_out/bench_r1_$tag.err
python - <<P
import json
d=json.load(open("gpurun_out/bench_r1_$tag.json"))
print("ms_per_step", d["ms_per_step"])
print({k: round(v, 3) for k, v in d["phase_ms"].items()})
r=d["roofline"]
if r:
    print("dominant", r["kernel"], "GB/s", round(r["achieved"]), "launches", r["launches"], "avg ms", r["avg_launch_ms"])
    print({k: (round(v["ms_per_step"], 3), int(v["launches_per_step"])) for k, v in sorted(r["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"])})
P
