import sys, time
sys.path.insert(0, '.')
import numpy as np
from cairo_m_amd import Backend
from cairo_m_amd.lib import synth_fibonacci
be = Backend(0)
inps = [synth_fibonacci(n) for n in (5, 3000, 100000, 419000)]
devs = [be.upload_input(i) for i in inps]
alone = []
for d in devs:
    p = be.prove_device(d); alone.append(p.words().copy()); p.free()
rng = np.random.default_rng(0)
bad = 0
for rnd in range(4):
    order = list(rng.integers(0, len(devs), size=24))
    t = time.perf_counter()
    proofs = be.prove_many([devs[k] for k in order], inflight=8)
    dt = time.perf_counter() - t
    for k, p in zip(order, proofs):
        if not np.array_equal(p.words(), alone[k]): bad += 1
        if rnd == 0 and p.verify()[0] != 0: bad += 1
        p.free()
    print('round', rnd, 'ms', round(dt*1e3, 1), 'bad', bad)
assert bad == 0
print('stress ok')
