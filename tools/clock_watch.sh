#!/bin/bash
# Sustained shader clock / power while proofs run back to back (is the chip power-limited under this load?):
#   tools/clock_watch.sh <tag> [inflight]  -> gpurun_out/<tag>_clocks.txt  (rocm-smi samples next to a 12-second proving loop)
tag=${1:-x}; inflight=${2:-1}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${tag}_clocks.txt
rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|mclk|Power|busy" > $out
python - "$inflight" >> $out 2>&1 <<'P' &
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from cairo_m_amd.lib import Backend, synth_fibonacci
inflight = int(sys.argv[1])
be = Backend(0)
inp = synth_fibonacci(419000)
dev = be.upload_input(inp)
for _ in range(3): be.prove_device(dev).free()
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 12:
    if inflight > 1:
        ps = be.prove_many([dev] * 16, inflight=inflight)
        for p in ps: p.free()
        n += 16
    else:
        be.prove_device(dev).free(); n += 1
torch.cuda.synchronize()
print(f"loop: {n} proofs, {(time.perf_counter() - t0) / n * 1e3:.3f} ms per proof (inflight {inflight})")
P
sleep 6
for i in 1 2 3 4 5; do rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|Power|busy" | tr '\n' ' ' >> $out; echo >> $out; sleep 1; done
wait
cat $out
