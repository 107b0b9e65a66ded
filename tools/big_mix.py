import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from cairo_m_amd import Backend
from cairo_m_amd.workloads import all_opcodes_program
from cairo_m_amd.lib import vm_segment
iters = int(sys.argv[1])
be = Backend(0)
t = time.time(); prog, steps = all_opcodes_program(iters); print("program", len(prog), "steps", steps, f"{time.time()-t:.1f}s", flush=True)
t = time.time(); hs = vm_segment(prog, entry_pc=0, args=(), n_returns=0); print(f"vm {time.time()-t:.1f}s", flush=True)
t = time.time(); dev = be.adapt_segment(hs); torch.cuda.synchronize(); print(f"adapt {time.time()-t:.2f}s", flush=True)
for i in range(2):
    t = time.time(); p = be.prove_device(dev); torch.cuda.synchronize(); dt = time.time()-t
    st = p.stats(); print(f"prove {dt*1e3:.1f} ms cells {st['cells']:.3e} cells/s {st['cells']/dt:.3e}", flush=True)
    free, total = torch.cuda.mem_get_info(); print(f"hbm used {(total-free)/2**30:.1f} GiB of {total/2**30:.0f}", flush=True)
    if i == 0: p.free()
t = time.time(); rc, err = p.verify(); print("verify", rc, err, f"{time.time()-t:.2f}s")
