"""Intra-proof sharding (BASELINE configs[3]): 2 and 4 ranks prove ONE segment — components split across
the ranks for trace / LogUp / IFFT / LDE / constraints / OODS, rows split for Merkle hashing and DEEP quotients, collectives
through cm_comm — and every rank's proof is bit-identical to the single-GPU proof of the same input.  The ranks share the
one GPU of the test box and talk over gloo (the multi-GPU path is the same code with backend nccl = RCCL)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from cairo_m_amd.lib import synth_fibonacci

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_sharded(world, fib_n, out, mixed_iters=0, comm="torch", extra_env=None, cfg=None):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "-m", "cairo_m_amd.sharded", "--fib-n", str(fib_n), "--dist-backend", "gloo",
           "--force-device", "0", "--steps", "0", "--out", out, "--mixed-iters", str(mixed_iters), "--comm", comm]
    if cfg:
        cmd += ["--cfg", ",".join(str(x) for x in cfg)]
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(extra_env or {})
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout


@pytest.mark.parametrize("world,fib_n", [(2, 50), (2, 30_000), (4, 2_000), (8, 419_000), (2, 1_677_000)])
def test_sharded_proof_equals_single_gpu_proof(backend, oracle, tmp_path, world, fib_n):
    # (8, 419 000): the metric config over 8 ranks; (2, 1 677 000): BASELINE configs[3]'s 2^24-row segment (7.7e8 cells)
    inp = synth_fibonacci(fib_n)
    p = backend.prove(inp)
    want = p.words().copy()
    p.free()
    out = str(tmp_path / "proof")
    log = _run_sharded(world, fib_n, out)
    for r in range(world):
        got = np.load(f"{out}.{r}.npy")
        assert got.size == want.size, (r, got.size, want.size, log[-500:])
        diff = np.nonzero(got != want)[0]
        assert diff.size == 0, f"rank {r}: first differing words {diff[:8]} of {got.size}"
    assert oracle.verify(want)[0] == 0
    inp.free()


@pytest.mark.parametrize("world,fib_n,stop", [(2, 3_000, 7), (4, 30_000, 8), (8, 30_000, 9), (2, 30_000, 99)])
def test_sharded_fri_layers_by_row_range(backend, oracle, tmp_path, world, fib_n, stop):
    """FRI is committed by ROW RANGE above 2^stop rows (first-layer tree over the quotient slices, inner layers folded slice to
    slice, one 32-byte all-gather per tree), then gathered and finished on every rank.  The default stop is 2^16; here it is
    lowered (CM_SHARD_FRI_STOP_LOG) so that small proofs run many sharded layers, and set to 99 for the fully replicated FRI of
    the earlier rounds: the proof never changes."""
    inp = synth_fibonacci(fib_n)
    p = backend.prove(inp)
    want = p.words().copy()
    p.free()
    out = str(tmp_path / "proof")
    log = _run_sharded(world, fib_n, out, extra_env={"CM_SHARD_FRI_STOP_LOG": str(stop)})
    for r in range(world):
        got = np.load(f"{out}.{r}.npy")
        assert got.size == want.size, (r, got.size, want.size, log[-500:])
        diff = np.nonzero(got != want)[0]
        assert diff.size == 0, f"rank {r}: first differing words {diff[:8]} of {got.size}"
    inp.free()


@pytest.mark.parametrize("world,fib_n,env", [
    (2, 30_000, {"CM_SHARD_TREE_STREAM": "0"}),
    (4, 3_000, {"CM_SHARD_TREE_STREAM": "0", "CM_SHARD_FRI_STREAM": "0", "CM_SHARD_SPLIT_MIN_LOG": "6", "CM_SHARD_FRI_STOP_LOG": "8"}),
    (2, 30_000, {"CM_SHARD_FRI_STREAM": "0", "CM_SHARD_FRI_STOP_LOG": "9"}),
    (8, 30_000, {"CM_SHARD_TREE_STREAM": "1", "CM_SHARD_SPLIT_MIN_LOG": "7", "CM_SHARD_FRI_STOP_LOG": "9", "CM_QUOT_LEAF": "0"}),
])
def test_sharded_transcript_forms(backend, oracle, tmp_path, world, fib_n, env):
    """The transcript steps of the sharded prover either stay on the stream (device tree top + the single-GPU prover's step
    kernels, host replay at the next wait: "shard_tree_stream" for the four commitment trees, "shard_fri_stream" for the FRI layers;
    the default) or go through the host after every tree (0).  Every combination gives the single-GPU proof."""
    inp = synth_fibonacci(fib_n)
    p = backend.prove(inp)
    want = p.words().copy()
    p.free()
    out = str(tmp_path / "proof")
    log = _run_sharded(world, fib_n, out, extra_env=env)
    for r in range(world):
        got = np.load(f"{out}.{r}.npy")
        assert got.size == want.size, (r, got.size, want.size, log[-500:])
        diff = np.nonzero(got != want)[0]
        assert diff.size == 0, f"rank {r}: first differing words {diff[:8]} of {got.size}"
    inp.free()


@pytest.mark.parametrize("world,fib_n,min_log", [(2, 200, 5), (4, 3_000, 6), (8, 30_000, 7), (4, 100_000, 12), (2, 30_000, 99)])
def test_split_components_by_rows_and_columns(backend, oracle, tmp_path, world, fib_n, min_log):
    """Large opcode components are SPLIT over the ranks (ShardPlan): trace rows, lookup histograms, LogUp rows and constraint
    quotients by row range, IFFT / LDE / OODS by column, with the rows -> columns transposes in between and the cumulative-sum
    LDE columns gathered for the previous-row mask.  CM_SHARD_SPLIT_MIN_LOG lowers the size threshold so that small proofs split
    most of their opcode components (down to 4-row slices); 99 = never split (the whole-component plan of rounds 2-3).  The proof
    never changes."""
    inp = synth_fibonacci(fib_n)
    p = backend.prove(inp)
    want = p.words().copy()
    p.free()
    out = str(tmp_path / "proof")
    log = _run_sharded(world, fib_n, out, extra_env={"CM_SHARD_SPLIT_MIN_LOG": str(min_log)})
    for r in range(world):
        got = np.load(f"{out}.{r}.npy")
        assert got.size == want.size, (r, got.size, want.size, log[-500:])
        diff = np.nonzero(got != want)[0]
        assert diff.size == 0, f"rank {r}: first differing words {diff[:8]} of {got.size}"
    assert oracle.verify(want)[0] == 0
    inp.free()


@pytest.mark.parametrize("world,fib_n,cfg", [(2, 300, (10, 2, 0, 30)), (4, 30_000, (8, 3, 1, 20)), (8, 3_000, (8, 2, 2, 16))])
def test_sharded_log_blowup_factor_above_one(backend, oracle, tmp_path, world, fib_n, cfg):
    """log_blowup_factor > 1 in the sharded prover: commitment domains of log + B, constraints on the (log + 1) domain of every
    owner's own polynomials (whole components: no split), FRI with the larger blowup — equal to the single-GPU proof under the same
    PcsConfig, which the oracle's verifier accepts under that config."""
    inp = synth_fibonacci(fib_n)
    p = backend.prove(inp, cfg=cfg)
    want = p.words().copy()
    p.free()
    out = str(tmp_path / "proof")
    log = _run_sharded(world, fib_n, out, cfg=cfg)
    for r in range(world):
        got = np.load(f"{out}.{r}.npy")
        assert got.size == want.size, (r, got.size, want.size, log[-500:])
        diff = np.nonzero(got != want)[0]
        assert diff.size == 0, f"rank {r}: first differing words {diff[:8]} of {got.size}"
    assert oracle.verify(want, cfg=cfg)[0] == 0
    inp.free()


@pytest.mark.parametrize("world,iters", [(2, 300), (4, 24_000), (8, 100_000)])
def test_sharded_all_opcode_proof_equals_single_gpu_proof(backend, oracle, tmp_path, world, iters):
    """BASELINE configs[4] shape: every opcode component live (24 of them with ~iters rows each), so the ownership plan spreads
    many equal-size components over the ranks instead of fibonacci's five; (8, 100 000) is 4.35 M steps."""
    from cairo_m_amd.lib import vm_run
    from cairo_m_amd.workloads import all_opcodes_program
    inp = vm_run(all_opcodes_program(iters)[0], entry_pc=0, args=(), n_returns=0)
    p = backend.prove(inp)
    want = p.words().copy()
    p.free()
    out = str(tmp_path / "proof")
    log = _run_sharded(world, 0, out, mixed_iters=iters)
    for r in range(world):
        got = np.load(f"{out}.{r}.npy")
        assert got.size == want.size, (r, got.size, want.size, log[-500:])
        diff = np.nonzero(got != want)[0]
        assert diff.size == 0, f"rank {r}: first differing words {diff[:8]} of {got.size}"
    assert oracle.verify(want)[0] == 0
    inp.free()


@pytest.mark.parametrize("fib_n", [50, 100_000])
def test_in_library_rccl_comm_world_1(backend, oracle, tmp_path, fib_n):
    """The library's own stream-ordered RCCL communicator (cm_rccl_comm_create: ncclAllGather / grouped ncclSend + ncclRecv on
    the prover's stream, no host synchronisation): the test box has ONE GPU and RCCL refuses two ranks on one device, so this
    runs the sharded path with world = 1 — the device collectives a lone rank still issues (sub-root, histogram, coefficient
    and sampled-value all-gathers go through ncclAllGather; the row exchange and the host-word gathers are short-cut for a rank
    that owns everything) — and requires the proof to equal the single-GPU one.  (Multi-rank bit-exactness is covered over gloo above; multi-rank RCCL needs a multi-GPU node.)"""
    inp = synth_fibonacci(fib_n)
    p = backend.prove(inp)
    want = p.words().copy()
    p.free()
    out = str(tmp_path / "proof")
    _run_sharded(1, fib_n, out, comm="rccl")
    got = np.load(f"{out}.0.npy")
    assert got.size == want.size and np.array_equal(got, want)
    inp.free()


def test_failing_rank_calls_the_communicator_abort(backend):
    """cm_comm::abort: a rank that fails inside cm_prove_sharded (here: its all_gather callback reports an error) tells the
    communicator before the error is returned, so that the peers' next collective does not wait for it for ever."""
    import ctypes as C
    from cairo_m_amd.sharded import CmComm, _A2A, _AG, _SETSTREAM, _ABORT, shard_plan
    inp = synth_fibonacci(50)
    _, words = shard_plan(inp, 1, backend.L)
    send, recv = backend.col_alloc(words), backend.col_alloc(words)   # cm_handle = the device address of the words
    aborted = []
    a2a = _A2A(lambda ctx, s, r: 1)
    ag = _AG(lambda ctx, w: 1)
    ab = _ABORT(lambda ctx: aborted.append(True))
    comm = CmComm(0, 1, None, int(send), int(recv), words, a2a, ag, 0, _SETSTREAM(), ab)
    dev = backend.upload_input(inp)
    out = C.c_void_p()
    rc = backend.L.cm_prove_sharded(dev, None, C.byref(comm), C.byref(out))
    assert rc != 0 and aborted == [True], (rc, aborted)
    buf = C.create_string_buffer(1024)
    backend.L.cm_last_error(buf, C.c_size_t(1024))
    assert b"callback failed" in buf.value, buf.value
    backend.free_input(dev)
    backend.col_free(send)
    backend.col_free(recv)
    inp.free()


def _gpus():
    import torch
    return torch.cuda.device_count()


@pytest.mark.parametrize("world,fib_n,comm", [(2, 50, "rccl"), (2, 419_000, "rccl"), (2, 419_000, "torch"), (4, 30_000, "rccl"),
                                               (8, 419_000, "rccl")])
def test_multi_gpu_rccl_proof_equals_single_gpu_proof(backend, oracle, tmp_path, world, fib_n, comm):
    """Switches itself on where the box has `world` GPUs (the round's test boxes have one: skipped there).  One process per GPU
    (LOCAL_RANK = device), backend nccl = RCCL over xGMI: the library's own stream-ordered communicator (cm_rccl_comm_create —
    ncclAllGather and the grouped ncclSend / ncclRecv row exchange of comm_rccl.hip get real peers here for the first time) or the
    torch.distributed callbacks, proving the metric config as ONE sharded proof.  Every rank's words must equal the single-GPU
    proof's; --check-single makes each rank also compare against its own single-GPU proof.  This is the first thing to run on a
    multi-GPU node, before `bench.py --gpus N`."""
    if _gpus() < world:
        pytest.skip(f"needs {world} GPUs, this box has {_gpus()}")
    inp = synth_fibonacci(fib_n)
    p = backend.prove(inp)
    want = p.words().copy()
    p.free()
    out = str(tmp_path / "proof")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "-m", "cairo_m_amd.sharded", "--fib-n", str(fib_n), "--dist-backend", "nccl",
           "--comm", comm, "--steps", "1", "--check-single", "--json", "--out", out]
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    import json
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["world"] == world and line["bit_identical_to_single_gpu_proof"] is True, line
    for k in range(world):
        got = np.load(f"{out}.{k}.npy")
        assert got.size == want.size and np.array_equal(got, want), f"rank {k} differs from the single-GPU proof"
    assert oracle.verify(want)[0] == 0
    inp.free()


def test_bench_multi_gpu_replicas_line(tmp_path):
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one rank per GPU, RCCL barrier + max over ranks):
    runs only where the box has two GPUs; the line must report n_gpus = 2, weak scaling and about twice one GPU's cells per proof
    batch (every rank proves its own segment replica)."""
    if _gpus() < 2:
        pytest.skip(f"needs 2 GPUs, this box has {_gpus()}")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline"]
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    import json
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
