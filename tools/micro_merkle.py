"""Micro-benchmark: Merkle commit of a 4-column 2^22 tree (FRI-layer shape) — used under rocprofv3 --pmc."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cairo_m_amd import Backend
be = Backend(0)
log = int(sys.argv[1]) if len(sys.argv) > 1 else 22
ncols = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rng = np.random.default_rng(0)
cols = [be.upload(rng.integers(0, 2**31 - 1, size=1 << log, dtype=np.uint32)) for _ in range(ncols)]
for it in range(3):
    t = time.perf_counter()
    root = be.merkle_commit(cols, [log] * ncols)
    print("commit ms", (time.perf_counter() - t) * 1e3)
