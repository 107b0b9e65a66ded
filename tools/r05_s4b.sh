#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
cat > /tmp/one.py <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from cairo_m_amd.lib import Backend, synth_fibonacci
be = Backend(0)
inp = synth_fibonacci(3)
try:
    p = be.prove(inp); print("ok", p.words().size)
except Exception as e:
    print("FAIL", e)
PY
for m in 0 1 2; do
  echo "== CM_FLAG_MEM=$m"; CM_FLAG_MEM=$m CM_JOIN_LIMIT_S=1 CM_FLAG_JOIN_DEBUG=1 timeout 100 python /tmp/one.py 2>&1 | tail -12
done
