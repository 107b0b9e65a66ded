"""Multi-rank plumbing on CPU (gloo, world_size 2, no GPU):
* the sharded prover's host side — every rank derives the SAME component -> rank plan (cm_shard_plan is host code) and the
  two cm_comm collectives (cairo_m_amd/sharded.py::TorchComm) move exactly the rank-major blocks the library expects;
* replicas over segments: each rank checks the AIR of its own continuation segment with the CPU oracle."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import ctypes as C, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["CM_ROOT"])
from cairo_m_amd.lib import synth_fibonacci
from cairo_m_amd.sharded import TorchComm, shard_plan
from tests.oracle_binding import Oracle
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
# ---- plan: identical on every rank, every component owned, the two biggest components on different ranks
inp = synth_fibonacci(3000)
owner, words = shard_plan(inp, world)
t = torch.tensor(owner, dtype=torch.int32)
ref = t.clone(); dist.broadcast(ref, 0)
assert torch.equal(t, ref) and set(o for o in owner if o >= 0) == set(range(world)) and words > 0   # (-1: split over all ranks)
big, _ = shard_plan(synth_fibonacci(100_000), world)
assert big[7] == -1 and big[6] == -1  # store_fp_imm and store_fp_fp, the two heaviest components of a large fibonacci_loop: split over the ranks
# ---- collectives on CPU staging buffers (the GPU path differs only in where the buffers live)
comm = TorchComm(1 << 12)
send = np.frombuffer((C.c_uint32 * (1 << 12)).from_address(comm.send.data_ptr()), dtype=np.uint32)
recv = np.frombuffer((C.c_uint32 * (1 << 12)).from_address(comm.recv.data_ptr()), dtype=np.uint32)
# all_gather: block r of every recv buffer = rank r's send block
send[:5] = 100 * rank + np.arange(5)
assert comm.c.all_gather(None, 5) == 0
assert recv[:10].tolist() == list(range(0, 5)) + list(range(100, 105))
# all_to_all_v with unequal splits: rank r sends (r + 1) words to rank 0 and 2 * (r + 1) words to rank 1
sw = [(rank + 1), 2 * (rank + 1)]
send[:sum(sw)] = 1000 * rank + np.arange(sum(sw))
rw = [(s + 1) * (1 if rank == 0 else 2) for s in range(world)]
assert comm.c.all_to_all_v(None, (C.c_uint64 * 2)(*sw), (C.c_uint64 * 2)(*rw)) == 0
want = []
for s in range(world):
    off = 0 if rank == 0 else (s + 1)
    want += [1000 * s + off + k for k in range(rw[s])]
assert recv[:sum(rw)].tolist() == want, (rank, recv[:sum(rw)].tolist(), want)
inp.free()
# ---- replicas: rank r checks continuation segment r with the oracle
orc = Oracle(os.path.join(os.environ["CM_ROOT"], "oracle", "liboracle.so"))
seg = synth_fibonacci(30, max_steps=200, segment=rank)
rc, err = orc.assert_constraints(seg.view)
assert rc == 0, err
ok = torch.tensor([1]); dist.all_reduce(ok)
assert ok.item() == world
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_two_ranks_gloo(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, CM_ROOT=ROOT, PYTHONPATH=ROOT, OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(script)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("ok") == 2
