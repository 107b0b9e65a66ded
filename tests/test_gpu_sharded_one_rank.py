"""cm_prove_sharded with ONE rank inside the test process: the in-library stream-ordered RCCL communicator created
from a locally generated id (world = 1: no torch.distributed, no child process), so that the sharded prover's host code — the
device-side transcript steps and their host replays, the claimed sums' device gather, the FRI hand-over, the two pipelined
decommitment gathers and the proof assembly — also runs under the host-ASAN build (tools/asan_run.sh keeps the multi-process
sharded tests out: torch's child processes do not start under a preloaded sanitizer runtime).  Every proof equals the
single-GPU proof word for word, in every form of the sharded prover's switches."""
import ctypes as C

import numpy as np
import pytest

from cairo_m_amd.lib import synth_fibonacci
from cairo_m_amd.sharded import RcclComm, prove_sharded, shard_plan

pytestmark = pytest.mark.gpu


def _comm(backend, inp, cfg=None):
    _, words = shard_plan(inp, 1, backend.L, cfg)
    idb = (C.c_uint8 * 128)()
    backend._ck(backend.L.cm_rccl_unique_id(idb))
    return RcclComm(backend, words, rank=0, world=1, id_bytes=bytes(idb))


def _set(backend, key, v):
    assert backend.L.cm_set_tuning(key.encode(), C.c_int32(v)) == 0, key


DEFAULTS = {"shard_tree_stream": 1, "shard_fri_stream": 1, "shard_fri_stop_log": 16, "quot_leaf": 1, "fri_fold_leaf": 1, "shard_halo": 1}


@pytest.mark.parametrize("fib_n,cfg", [(50, None), (3_000, None), (100_000, None), (30_000, (8, 2, 1, 20)), (419_000, None)])
def test_one_rank_proof_equals_single_gpu_proof(backend, oracle, fib_n, cfg):
    inp = synth_fibonacci(fib_n)
    dev = backend.upload_input(inp)
    p = backend.prove_device(dev, cfg)
    want = p.words().copy()
    p.free()
    comm = _comm(backend, inp, cfg)
    try:
        for _ in range(2):   # (the second proof runs on a warm pool and behind the first one's teardown)
            q = prove_sharded(backend, dev, comm, cfg)
            got = q.words().copy()
            q.free()
            assert got.size == want.size and np.array_equal(got, want)
        if fib_n <= 3_000:
            assert oracle.verify(want, cfg)[0] == 0
    finally:
        comm.free()
        backend.free_input(dev)
        inp.free()


@pytest.mark.parametrize("switches", [
    {"shard_tree_stream": 0},
    {"shard_fri_stream": 0},
    {"shard_tree_stream": 0, "shard_fri_stream": 0, "shard_fri_stop_log": 9},
    {"shard_fri_stop_log": 8},
    {"shard_fri_stop_log": 99},
    {"quot_leaf": 0, "fri_fold_leaf": 0},
])
def test_one_rank_switch_forms_keep_the_proof_bytes(backend, switches):
    inp = synth_fibonacci(100_000)
    dev = backend.upload_input(inp)
    p = backend.prove_device(dev)
    want = p.words().copy()
    p.free()
    comm = _comm(backend, inp)
    try:
        for k, v in switches.items():
            _set(backend, k, v)
        q = prove_sharded(backend, dev, comm)
        got = q.words().copy()
        q.free()
        assert got.size == want.size and np.array_equal(got, want), switches
    finally:
        for k in switches:
            _set(backend, k, DEFAULTS[k])
        comm.free()
        backend.free_input(dev)
        inp.free()


def test_one_rank_all_opcode_segment(backend):
    """The all-opcode loop (every opcode component non-trivial, the builtins' tables busy) through the sharded prover."""
    from cairo_m_amd.lib import vm_run
    from cairo_m_amd.workloads import all_opcodes_program
    inp = vm_run(all_opcodes_program(3_000)[0], entry_pc=0, args=(), n_returns=0)
    dev = backend.upload_input(inp)
    p = backend.prove_device(dev)
    want = p.words().copy()
    p.free()
    comm = _comm(backend, inp)
    try:
        q = prove_sharded(backend, dev, comm)
        got = q.words().copy()
        q.free()
        assert got.size == want.size and np.array_equal(got, want)
    finally:
        comm.free()
        backend.free_input(dev)
        inp.free()
