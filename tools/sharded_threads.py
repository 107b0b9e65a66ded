# development tool: cm_prove_sharded with N ranks as THREADS of one process on ONE GPU over the stream-ordered loop-back
# communicator of tests/test_gpu_sharded_threads.py (the production code path of a multi-GPU node, minus RCCL and minus the extra
# GPUs): the time per proof is the SUM of all ranks' work on one device + every exchange as device copies — what sharding ADDS in
# total GPU work and host time, not a scaling figure.   python tools/sharded_threads.py [--fib-n 419000] [--worlds 1,2,4,8] [--reps 8]
import argparse, ctypes as C, os, statistics, sys, threading, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from cairo_m_amd.lib import Backend, Proof, synth_fibonacci
from cairo_m_amd.sharded import shard_plan
from tests.test_gpu_sharded_threads import Loopback, LoopbackRank

ap = argparse.ArgumentParser()
ap.add_argument("--fib-n", type=int, default=419000)
ap.add_argument("--worlds", default="1,2,4,8")
ap.add_argument("--reps", type=int, default=8)
a = ap.parse_args()
be = Backend(0)
inp = synth_fibonacci(a.fib_n)
dev = be.upload_input(inp)
p = be.prove_device(dev); want = p.words().copy(); p.free()
one = []
for _ in range(a.reps + 3):
    t = time.perf_counter(); be.prove_device(dev).free(); one.append((time.perf_counter() - t) * 1e3)
print(f"single-GPU prover: {statistics.median(one[3:]):.2f} ms per proof")
for w in [int(x) for x in a.worlds.split(",")]:
    _, words = shard_plan(inp, w, be.L, None)
    shared = Loopback(be, w, words)
    ranks = [LoopbackRank(shared, r) for r in range(w)]
    bar = threading.Barrier(w + 1)
    times, ok, phases = [], [True], [None]

    def work(r):
        for k in range(a.reps + 3):
            bar.wait()
            h = C.c_void_p()
            rc = be.L.cm_prove_sharded(dev, None, C.byref(ranks[r].c), C.byref(h))
            assert rc == 0, rc
            pr = Proof(be.L, h)
            if k == a.reps + 2:
                ok[0] = ok[0] and bool(np.array_equal(pr.words(), want))
                if r == 0:
                    phases[0] = {n: round(v, 2) for n, v in pr.stats()["phase_ms"].items()}
            pr.free()
            bar.wait()
    ts = [threading.Thread(target=work, args=(r,)) for r in range(w)]
    for t in ts:
        t.start()
    for k in range(a.reps + 3):
        bar.wait(); t0 = time.perf_counter()
        bar.wait(); times.append((time.perf_counter() - t0) * 1e3)
    for t in ts:
        t.join()
    calls = shared.calls // (a.reps + 3)
    shared.free()
    print(f"{w} rank(s) as threads on one GPU: {statistics.median(times[3:]):.2f} ms per proof, {calls} collectives per proof, bit-identical {ok[0]}; rank 0 phases {phases[0]}")
