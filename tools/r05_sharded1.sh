#!/bin/bash
# sharded prover with world = 1 (the single-rank tax): phase times next to the single-GPU prover's
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for comm in rccl torch; do
  echo "== comm $comm"
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 -m cairo_m_amd.sharded --fib-n 419000 --steps 4 --comm $comm 2>&1 | tail -2
done
python - <<'PY'
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from cairo_m_amd.lib import Backend, synth_fibonacci
be = Backend(0); inp = synth_fibonacci(419000); dev = be.upload_input(inp)
for _ in range(3): be.prove_device(dev).free()
p = be.prove_device(dev); print("single:", {k: round(v, 3) for k, v in p.stats()["phase_ms"].items()}); p.free()
PY
