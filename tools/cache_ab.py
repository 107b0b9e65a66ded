import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from cairo_m_amd.lib import Backend, synth_fibonacci
be = Backend(0)
dev = be.upload_input(synth_fibonacci(419000))
def run(n=10):
    for _ in range(3): be.prove_device(dev).free()
    torch.cuda.synchronize(); t=time.perf_counter()
    ph={}
    for _ in range(n):
        p=be.prove_device(dev)
        for k,v in p.stats()["phase_ms"].items(): ph[k]=ph.get(k,0)+v/n
        p.free()
    torch.cuda.synchronize()
    return (time.perf_counter()-t)/n*1e3, {k:round(v,2) for k,v in ph.items()}
for pp,tw in ((0,0),(1,0),(0,1),(1,1),(0,0)):
    be.set_preprocessed_cache(bool(pp)); be.set_twiddle_cache(bool(tw))
    ms,ph=run()
    print(f"pp_cache {pp} tw_cache {tw}: {ms:.3f} ms", ph)
