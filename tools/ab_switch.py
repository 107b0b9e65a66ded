# development tool: two or more forms of a cm_set_tuning switch timed ALTERNATELY inside one process (boxes differ by +-4 %, runs of
# one process by +-2 %; alternating blocks of lone proofs on one warm process removes both):
#   python tools/ab_switch.py [--reps 12] [--block 8] oods_poll stage_copy_kernel oods_split=650,780
# (key alone: 1 against 0; key=a,b: value a against value b)
# prints, per key, the median ms per proof with the switch on and off (all other switches at their defaults) and the paired difference.
import argparse, os, statistics, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ctypes as C
import torch
from cairo_m_amd.lib import Backend, synth_fibonacci

ap = argparse.ArgumentParser()
ap.add_argument("keys", nargs="+")
ap.add_argument("--reps", type=int, default=12)
ap.add_argument("--block", type=int, default=8)
ap.add_argument("--fib-n", type=int, default=419000)
ap.add_argument("--phases", action="store_true", help="also the mean GPU phase times per form")
a = ap.parse_args()
be = Backend(0)
dev = be.upload_input(synth_fibonacci(a.fib_n))


PH = {}
DEFAULTS = [("oods_split", 780), ("merkle_npw", -1), ("fork_width", 0), ("tree0_prio", -1), ("logup_width", 4), ("quot_rows", 2),
            ("fft_chunk_mb", 0), ("pace", -1), ("cons_plan", 0o01237456), ("logup_small_stream", -1), ("tail_grind_cap", 0), ("tw_batch", 8), ("fft_half_occ", 0), ("tree0_guest", 0), ("merkle_multi_top", 19)]   # every other switch defaults to 1


def block(tag=None):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(a.block):
        p = be.prove_device(dev)
        if tag is not None:   # GPU-side phase times (HIP events on the main stream), summed per variant
            for k, v in p.stats()["phase_ms"].items():
                PH.setdefault(tag, {}).setdefault(k, []).append(v)
        p.free()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / a.block * 1e3


def setk(key, v):
    if key == "cpu_affinity_sticky":   # cm_set_cpu_affinity: 2 = sticky, 1 = scoped (the default)
        assert be.L.cm_set_cpu_affinity(C.c_int32(2 if v else 1)) == 0
        return
    assert be.L.cm_set_tuning(key.encode(), C.c_int32(v)) == 0, key


for _ in range(2):
    block()
for spec in a.keys:
    key, _, vals = spec.partition("=")
    von, voff = (int(x, 0) for x in vals.split(",")) if vals else (1, 0)
    on, off = [], []
    for r in range(a.reps):
        order = (1, 0) if r % 2 == 0 else (0, 1)
        for v in order:
            setk(key, von if v else voff)
            block()                      # one untimed block after every flip
            (on if v else off).append(block((key, v)))
    setk(key, {e[0]: e[1] for e in DEFAULTS}.get(key, 1))
    d = [x - y for x, y in zip(on, off)]
    print(f"{spec:24s} on {statistics.median(on):.3f} ms  off {statistics.median(off):.3f} ms  paired on - off: median {statistics.median(d):+.3f}"
          f"  mean {statistics.mean(d):+.3f}  ({sum(1 for x in d if x < 0)} of {len(d)} pairs faster on)")
    if a.phases:
        for k in PH[(key, 1)]:
            m1, m0 = statistics.mean(PH[(key, 1)][k]), statistics.mean(PH[(key, 0)][k])
            print(f"    {k:22s} on {m1:7.3f}  off {m0:7.3f}  {m1 - m0:+.3f}")
