# development tool: the sharded prover on ONE rank (in-library RCCL communicator, world = 1) with forms of a cm_set_tuning switch timed
# ALTERNATELY inside one process, next to the single-GPU prover in the same process (boxes differ by +-4 %):
#   python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29517 tools/ab_sharded.py \
#       [--reps 8] [--block 6] shard_tree_stream shard_fri_stop_log=18,16
# (key alone: 1 against 0; key=a,b: value a against value b; no key: only the single-GPU / sharded pair)
import argparse, os, statistics, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ctypes as C
import torch
import torch.distributed as dist
from cairo_m_amd.lib import Backend, synth_fibonacci
from cairo_m_amd.sharded import RcclComm, prove_sharded, shard_plan

ap = argparse.ArgumentParser()
ap.add_argument("keys", nargs="*")
ap.add_argument("--reps", type=int, default=8)
ap.add_argument("--block", type=int, default=6)
ap.add_argument("--fib-n", type=int, default=419000)
ap.add_argument("--phases", action="store_true")
a = ap.parse_args()
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
be = Backend(0)
inp = synth_fibonacci(a.fib_n)
owner, words = shard_plan(inp, 1, be.L, None)
comm = RcclComm(be, words)
dev = be.upload_input(inp)
DEFAULTS = {"shard_fri_stop_log": 16}
PH = {}


def block(sharded=True, tag=None):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(a.block):
        p = prove_sharded(be, dev, comm, None) if sharded else be.prove_device(dev)
        if tag is not None:
            for k, v in p.stats()["phase_ms"].items():
                PH.setdefault(tag, {}).setdefault(k, []).append(v)
        p.free()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / a.block * 1e3


def setk(key, v):
    assert be.L.cm_set_tuning(key.encode(), C.c_int32(v)) == 0, key


for _ in range(2):
    block(True); block(False)
one, sh = [], []
for r in range(a.reps):
    for s in ((0, 1) if r % 2 == 0 else (1, 0)):
        block(bool(s))
        (sh if s else one).append(block(bool(s), ("sharded", s)))
d = [x - y for x, y in zip(sh, one)]
print(f"{'sharded vs single-GPU':24s} sharded {statistics.median(sh):.3f} ms  single {statistics.median(one):.3f} ms  paired difference: median {statistics.median(d):+.3f}  mean {statistics.mean(d):+.3f}")
if a.phases:
    for k in PH[("sharded", 1)]:
        m1 = statistics.mean(PH[("sharded", 1)][k]); m0 = statistics.mean(PH[("sharded", 0)].get(k, [0.0]))
        print(f"    {k:22s} sharded {m1:7.3f}  single {m0:7.3f}  {m1 - m0:+.3f}")
for spec in a.keys:
    key, _, vals = spec.partition("=")
    von, voff = (int(x, 0) for x in vals.split(",")) if vals else (1, 0)
    on, off = [], []
    for r in range(a.reps):
        for v in ((1, 0) if r % 2 == 0 else (0, 1)):
            setk(key, von if v else voff)
            block()
            (on if v else off).append(block(True, (key, v)))
    setk(key, DEFAULTS.get(key, 1))
    d = [x - y for x, y in zip(on, off)]
    print(f"{spec:24s} on {statistics.median(on):.3f} ms  off {statistics.median(off):.3f} ms  paired on - off: median {statistics.median(d):+.3f}"
          f"  mean {statistics.mean(d):+.3f}  ({sum(1 for x in d if x < 0)} of {len(d)} pairs faster on)")
    if a.phases:
        for k in PH[(key, 1)]:
            m1, m0 = statistics.mean(PH[(key, 1)][k]), statistics.mean(PH[(key, 0)][k])
            print(f"    {k:22s} on {m1:7.3f}  off {m0:7.3f}  {m1 - m0:+.3f}")
be.free_input(dev)
inp.free()
dist.destroy_process_group()
