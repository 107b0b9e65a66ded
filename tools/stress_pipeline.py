#!/usr/bin/env python3
"""Soak test of the concurrent paths of the library on one GPU (SURVEY §5 "race detection": the product uses LDS-cached atomics,
last-block tickets in k_merkle_top, 8 side streams + a pipelined commitment per proof, per-thread device pools, an upload ring and
pinned landing buffers; this drives all of them at once for a fixed time and requires every proof to be BIT-IDENTICAL to the proof
of the same input made alone).  Per round, until the time is up:

  1. cm_prove_many       24 resident segments of four sizes, 8 proofs in flight
  2. failure injection   the same with one or two inputs whose logged memory values break the constraints: status 10 must come
                         back, the bad slots must be empty, every other proof must be there and bit-identical, and the pipeline
                         must keep working
  3. cm_prove_many_host  streaming ingest of 12 host inputs, 4 in flight (uploads under proofs, inputs recycled through the pool)
  4. thread churn        3 NEW host threads (fresh thread-local streams, pools, pinned slots, ticket rings on recycled memory)
  5. framing freeze      cm_set_framing called every 2 ms while proofs run: refused whenever a prover is alive
  6. sharded, one rank   (round 6) two segments through cm_prove_sharded over the in-library RCCL communicator with world = 1 — the
                         device-side transcript steps and their host replays, the pipelined decommitment gathers — between the
                         pipelines above, on the pool and the pinned slots they leave behind

    python tools/stress_pipeline.py --minutes 10 > gpurun_out/<tag>_stress.txt      (one JSON summary line at the end)
"""
import argparse
import ctypes as C
import json
import sys
import threading
import time

sys.path.insert(0, '.')
import numpy as np
from cairo_m_amd import Backend
from cairo_m_amd.lib import CmError, ProverInputView, synth_fibonacci


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=2.0)
    ap.add_argument("--inflight", type=int, default=8)
    a = ap.parse_args()
    be = Backend(0)
    sizes = (5, 3000, 100000, 419000)
    inps = [synth_fibonacci(n) for n in sizes]
    devs = [be.upload_input(i) for i in inps]
    alone = []
    for d in devs:
        p = be.prove_device(d)
        alone.append(p.words().copy())
        assert p.verify()[0] == 0
        p.free()
    # a second copy of two inputs whose witness gets broken / repaired for the failure injection
    bad_inps = [synth_fibonacci(n) for n in (3000, 100000)]
    bad_of = {0: 1, 1: 2}     # bad_inps[i] has the size of inps[bad_of[i]]

    def flip(i):
        v = C.cast(bad_inps[i].view, C.POINTER(ProverInputView)).contents
        acc = np.ctypeslib.as_array(C.cast(v.data_accesses, C.POINTER(C.c_uint32)), shape=(int(v.n_data_accesses), 4))
        acc[20:40, 3] ^= 1
    from cairo_m_amd.sharded import RcclComm, prove_sharded, shard_plan
    idb = (C.c_uint8 * 128)()
    be._ck(be.L.cm_rccl_unique_id(idb))
    comm = RcclComm(be, max(shard_plan(i, 1, be.L, None)[1] for i in inps), rank=0, world=1, id_bytes=bytes(idb))
    rng = np.random.default_rng(0)
    stats = {"sharded_one_rank": 0, "rounds": 0, "proofs": 0, "mismatches": 0, "injected_failures": 0, "failures_reported": 0, "framing_refused": 0,
             "framing_slipped": 0, "streamed": 0, "churn_threads": 0}
    t_end = time.time() + 60 * a.minutes

    def mem_available_gb():
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) / 1048576.0
        return 0.0

    def rss_gb():
        for line in open("/proc/self/status"):
            if line.startswith("VmRSS:"):
                return int(line.split()[1]) / 1048576.0
        return 0.0
    vram0 = be.mem_info()[0] / 2**30
    stats["vram_free_gb_start"] = round(vram0, 1)
    mem0, rss0 = mem_available_gb(), rss_gb()
    stats["mem_available_gb_start"] = round(mem0, 1)

    def check(k, p):
        stats["proofs"] += 1
        if not np.array_equal(p.words(), alone[k]):
            stats["mismatches"] += 1
            print("MISMATCH size", sizes[k], flush=True)
        p.free()
    while time.time() < t_end:
        rnd = stats["rounds"]
        t0 = time.perf_counter()
        # 1. resident pipeline, with framing-change attempts from this thread while the workers prove
        order = [int(x) for x in rng.integers(0, len(devs), size=24)]
        out = {}
        th = threading.Thread(target=lambda: out.setdefault("p", be.prove_many([devs[k] for k in order], inflight=a.inflight)))
        th.start()
        while th.is_alive():
            # (the DEFAULT framing is what is "set": a call that slips in between two proofs must not change the proofs of this
            # soak — that a refused call leaves the setting alone and an accepted one takes effect is tests/test_gpu_framing.py's job)
            if be.L.cm_set_framing(b"") != 0:
                stats["framing_refused"] += 1
            else:
                stats["framing_slipped"] += 1
            time.sleep(0.002)
        th.join()
        assert be.L.cm_set_framing(b"") == 0
        for k, p in zip(order, out["p"]):
            check(k, p)
        # 2. failure injection through the streaming form (host inputs): 1-2 bad items among 10
        n_bad = 1 + rnd % 2
        for i in range(n_bad):
            flip(i)
        items = [("good", int(x)) for x in rng.integers(0, len(inps), size=10 - n_bad)] + [("bad", i) for i in range(n_bad)]
        rng.shuffle(items)
        stats["injected_failures"] += n_bad
        try:
            be.prove_many_host([inps[i] if kind == "good" else bad_inps[i] for kind, i in items], inflight=4)
            print("FAILURE NOT REPORTED", flush=True)
            stats["mismatches"] += 1
        except CmError as e:
            assert "status 10" in str(e), str(e)
            for (kind, i), p in zip(items, e.partial):
                if kind == "bad":
                    stats["failures_reported"] += 1 if p is None else 0
                    if p is not None:
                        stats["mismatches"] += 1
                        p.free()
                else:
                    assert p is not None
                    check(i, p)
        for i in range(n_bad):
            flip(i)                                   # repaired: the same buffers prove again
        for i, p in enumerate(be.prove_many_host(bad_inps[:n_bad], inflight=2)):
            check(bad_of[i], p)
        # 3. streaming ingest
        order = [int(x) for x in rng.integers(0, len(inps), size=12)]
        for k, p in zip(order, be.prove_many_host([inps[k] for k in order], inflight=4)):
            check(k, p)
            stats["streamed"] += 1
        # 4. thread churn (every fourth round: ~900 short-lived threads in ten minutes)
        res = {}

        def work(i, k):
            p = be.prove_device(devs[k])
            res[i] = (k, p)
        ks = [int(x) for x in rng.integers(0, len(devs), size=3)] if rnd % 4 == 0 else []
        ts = [threading.Thread(target=work, args=(i, k)) for i, k in enumerate(ks)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        for i, (k, p) in res.items():
            check(k, p)
        stats["churn_threads"] += len(ts)
        # 6. the sharded prover with one rank
        for k in [int(x) for x in rng.integers(0, len(devs), size=2)]:
            check(k, prove_sharded(be, devs[k], comm))
            stats["sharded_one_rank"] += 1
        stats["rounds"] += 1
        # a leak must end the soak, not the box: the thread-churn part once lost ~35 MB of pinned memory per thread (fixed:
        # engine.hpp at_thread_exit) and took the test box down after four minutes
        # (MemAvailable is reported, not acted on: the host is shared and it moves by tens of GB on its own)
        if rss_gb() > rss0 + 24 or be.mem_info()[0] / 2**30 < vram0 - 160:
            print(f"MEMORY GROWTH: MemAvailable {mem0:.1f} -> {mem_available_gb():.1f} GB, RSS {rss0:.1f} -> {rss_gb():.1f} GB, free VRAM "
                  f"{vram0:.1f} -> {be.mem_info()[0] / 2**30:.1f} GiB: stopping", flush=True)
            stats["mismatches"] += 1
            break
        print(f"round {rnd}: {time.perf_counter() - t0:.2f} s, proofs so far {stats['proofs']}, mismatches {stats['mismatches']}, free VRAM "
              f"{be.mem_info()[0] / 2**30:.1f} GiB, RSS {rss_gb():.1f} GB", flush=True)
    stats["minutes"] = a.minutes
    stats["mem_available_gb_end"] = round(mem_available_gb(), 1)
    stats["rss_growth_gb"] = round(rss_gb() - rss0, 2)
    stats["vram_free_gb_end"] = round(be.mem_info()[0] / 2**30, 1)
    stats["ok"] = stats["mismatches"] == 0 and stats["failures_reported"] == stats["injected_failures"]
    print(json.dumps(stats))
    if not stats["ok"]:
        sys.exit(1)


if __name__ == "__main__":
    main()
