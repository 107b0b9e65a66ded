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
# thread churn: every round starts NEW host threads (fresh thread-local streams, pools, pinned slots, ticket rings) on
# recycled device memory — a NULL-stream hipMemset of one of those once raced with the non-blocking prover streams
import threading
for rnd in range(6):
    res = {}
    def work(i, k):
        p = be.prove_device(devs[k]); res[i] = (k, p.words().copy()); p.free()
    ks = [int(x) for x in rng.integers(0, len(devs), size=3)]
    ts = [threading.Thread(target=work, args=(i, k)) for i, k in enumerate(ks)]
    for t in ts: t.start()
    for t in ts: t.join()
    for i, (k, w) in res.items():
        if not np.array_equal(w, alone[k]): bad += 1
    print('churn round', rnd, 'sizes', ks, 'bad', bad)
assert bad == 0
print('thread churn ok')
