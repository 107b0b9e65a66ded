# development tool: time a plain loop of lone proofs through the ctypes binding (no kprof events): python tools/lone_loop.py
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from cairo_m_amd.lib import Backend, synth_fibonacci
be = Backend(0)
inp = synth_fibonacci(419000)
dev = be.upload_input(inp)
for _ in range(3):
    be.prove_device(dev).free()
ts = []
for r in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(8):
        be.prove_device(dev).free()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t) / 8 * 1e3)
print("ms per proof:", [round(x, 3) for x in ts])
