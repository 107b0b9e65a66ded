"""cm_prove_sharded with a STREAM-ORDERED communicator and MORE THAN ONE rank on the one GPU of the test box.

RCCL refuses two ranks on one device, and the gloo runs of tests/test_gpu_sharded.py go through the host-blocking form of
cm_comm — so the code a multi-GPU node runs (CM_COMM_STREAM_ORDERED: collectives only enqueued on the prover's stream, the
claimed sums' device all-gather on the LogUp tail's side stream through cm_comm::set_stream, no host wait around an exchange)
would be exercised with one rank only.  Here the ranks are THREADS of this process, each with its own prover stream and staging
buffers, and the communicator is a loop-back: a collective is device-to-device copies from every rank's send buffer into the
caller's receive buffer, enqueued on the caller's current stream behind an event the owner recorded on ITS stream, plus a
second round of events so that nobody overwrites a send buffer a peer is still reading.  The host threads only meet at two
barriers per collective (to exchange the events): nothing waits for the GPU.  Every rank's proof equals the single-GPU proof."""
import ctypes as C
import threading

import numpy as np
import pytest

from cairo_m_amd.lib import Proof, synth_fibonacci
from cairo_m_amd.sharded import CmComm, _A2A, _AG, _SETSTREAM, _ABORT, shard_plan

pytestmark = pytest.mark.gpu
HIP = None   # libamdhip64, loaded by the first Loopback (not at collection: a box without a GPU only deselects this file)
hipMemcpyDeviceToDevice = 3
hipEventDisableTiming = 2


def _ck(rc, what):
    assert rc == 0, f"{what}: hip error {rc}"


class Loopback:
    """shared state of `world` loop-back ranks"""

    def __init__(self, backend, world, words):
        global HIP
        if HIP is None:
            HIP = C.CDLL("libamdhip64.so")
        self.world, self.words = world, words
        self.send = [backend.col_alloc(words) for _ in range(world)]
        self.recv = [backend.col_alloc(words) for _ in range(world)]
        self.stream = [C.c_void_p(0)] * world
        self.barrier = threading.Barrier(world)
        self.ready = [None] * world     # event of the current collective: rank's send buffer is complete
        self.done = [None] * world      # event: rank has enqueued (and, once it fires, finished) its reads of every send buffer
        self.arg = [None] * world       # per-rank argument of the current collective
        self.failed = False
        self.events = []
        self.lock = threading.Lock()
        self.calls = 0
        self.backend = backend

    def event(self):
        e = C.c_void_p()
        _ck(HIP.hipEventCreateWithFlags(C.byref(e), C.c_uint(hipEventDisableTiming)), "hipEventCreateWithFlags")
        with self.lock:
            self.events.append(e)
        return e

    def free(self):
        for e in self.events:
            HIP.hipEventDestroy(e)
        for h in self.send + self.recv:
            self.backend.col_free(h)


class LoopbackRank:
    def __init__(self, shared, rank):
        self.s, self.rank = shared, rank
        self._a2a, self._ag = _A2A(self._all_to_all_v), _AG(self._all_gather)
        self._ss, self._ab = _SETSTREAM(self._set_stream), _ABORT(self._abort)
        self.set_stream_calls = 0
        self.c = CmComm(rank, shared.world, None, int(shared.send[rank]), int(shared.recv[rank]), shared.words, self._a2a, self._ag,
                        1, self._ss, self._ab)   # flags = CM_COMM_STREAM_ORDERED

    def _set_stream(self, _ctx, stream):
        self.s.stream[self.rank] = C.c_void_p(stream)
        self.set_stream_calls += 1
        return 0

    def _abort(self, _ctx):
        self.s.failed = True
        self.s.barrier.abort()

    def _exchange(self, arg, copies_for):
        """copies_for(args of all ranks) -> [(source rank, source word offset, destination word offset, words)] of THIS rank"""
        s, r = self.s, self.rank
        st = s.stream[r]
        try:
            s.arg[r] = arg
            s.ready[r] = s.event()
            _ck(HIP.hipEventRecord(s.ready[r], st), "hipEventRecord")
            s.barrier.wait()                                    # every rank has recorded `ready` and published its argument
            for src, so, do, n in copies_for(list(s.arg)):
                if n == 0:
                    continue
                _ck(HIP.hipStreamWaitEvent(st, s.ready[src], C.c_uint(0)), "hipStreamWaitEvent")
                _ck(HIP.hipMemcpyAsync(C.c_void_p(int(s.recv[r]) + 4 * do), C.c_void_p(int(s.send[src]) + 4 * so), C.c_size_t(4 * n),
                                       C.c_int(hipMemcpyDeviceToDevice), st), "hipMemcpyAsync")
            s.done[r] = s.event()
            _ck(HIP.hipEventRecord(s.done[r], st), "hipEventRecord")
            s.barrier.wait()                                    # every rank has recorded `done`
            for peer in range(s.world):                         # my send buffer is free again once every peer has read it
                if peer != r:
                    _ck(HIP.hipStreamWaitEvent(st, s.done[peer], C.c_uint(0)), "hipStreamWaitEvent")
            s.barrier.wait()                                    # (nobody re-publishes ready / arg while a peer still reads them)
            if r == 0:
                s.calls += 1
            return 0
        except Exception as e:  # noqa: BLE001 — a Python exception must not unwind through the C frame
            print(f"[loop-back rank {r}] collective failed: {e!r}")
            return 1

    def _all_gather(self, _ctx, words_per_rank):
        w = int(words_per_rank)
        return self._exchange(w, lambda args: [(src, 0, src * w, w) for src in range(self.s.world)])

    def _all_to_all_v(self, _ctx, send_words, recv_words):
        n, r = self.s.world, self.rank
        sw = [int(send_words[i]) for i in range(n)]
        rw = [int(recv_words[i]) for i in range(n)]

        def copies(args):
            out, do = [], 0
            for src in range(n):
                ssw = args[src]                      # the source's send counts: its block for me starts behind its blocks for ranks < r
                assert ssw[r] == rw[src], (r, src, ssw, rw)
                out.append((src, sum(ssw[:r]), do, rw[src]))
                do += rw[src]
            return out
        return self._exchange(sw, copies)


def _prove_threads(backend, inp, world, cfg=None):
    _, words = shard_plan(inp, world, backend.L, cfg)
    shared = Loopback(backend, world, words)
    ranks = [LoopbackRank(shared, r) for r in range(world)]
    dev = backend.upload_input(inp)
    out, err = [None] * world, [None] * world
    cfg_arg = (C.c_uint32 * 4)(*cfg) if cfg else None

    def work(r):
        h = C.c_void_p()
        rc = backend.L.cm_prove_sharded(dev, cfg_arg, C.byref(ranks[r].c), C.byref(h))
        if rc != 0:
            buf = C.create_string_buffer(1024)
            backend.L.cm_last_error(buf, C.c_size_t(1024))
            err[r] = (rc, buf.value.decode(errors="replace"))
            return
        pr = Proof(backend.L, h)
        out[r] = pr.words()
        pr.free()
    ts = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    backend.free_input(dev)
    calls, set_streams = shared.calls, [x.set_stream_calls for x in ranks]
    shared.free()
    assert all(e is None for e in err), err
    return out, calls, set_streams


@pytest.mark.parametrize("world,fib_n", [(2, 3_000), (2, 30_000), (4, 30_000), (8, 100_000), (2, 419_000)])
def test_stream_ordered_ranks_as_threads(backend, world, fib_n):
    inp = synth_fibonacci(fib_n)
    p = backend.prove(inp)
    want = p.words().copy()
    p.free()
    got, calls, set_streams = _prove_threads(backend, inp, world)
    for r in range(world):
        assert got[r].size == want.size and np.array_equal(got[r], want), f"rank {r}"
    assert calls >= 8                                   # sub-root gathers, exchanges, sums, coefficients, samples, decommitment
    assert all(n >= 3 for n in set_streams), set_streams   # bound at the start, moved to the LogUp tail's stream and back
    inp.free()


DEFAULTS = {"shard_tree_stream": 1, "shard_fri_stream": 1, "shard_fri_stop_log": 16, "shard_halo": 1, "logup_defer": 1, "quot_leaf": 1,
            "fri_fold_leaf": 1}


@pytest.mark.parametrize("world,fib_n,cfg,switches", [
    (4, 30_000, None, {"shard_fri_stop_log": 8}),                       # many row-sharded FRI layers, slices of 2^6 rows
    (2, 30_000, None, {"shard_tree_stream": 0}),                        # host-driven transcript steps over an ordered communicator
    (2, 30_000, None, {"shard_fri_stream": 0, "shard_fri_stop_log": 9}),
    (4, 30_000, None, {"shard_halo": 0}),                               # cumulative-sum columns all-gathered
    (2, 100_000, None, {"logup_defer": 0, "quot_leaf": 0, "fri_fold_leaf": 0}),
    (2, 30_000, (8, 2, 1, 20), {}),                                     # log_blowup_factor 2: whole components, own (log + 1) domains
    (8, 30_000, None, {"shard_fri_stop_log": 99}),                      # FRI replicated
])
def test_stream_ordered_ranks_switch_forms(backend, world, fib_n, cfg, switches):
    inp = synth_fibonacci(fib_n)
    dev = backend.upload_input(inp)
    p = backend.prove_device(dev, cfg)
    want = p.words().copy()
    p.free()
    backend.free_input(dev)
    try:
        for k, v in switches.items():
            assert backend.L.cm_set_tuning(k.encode(), C.c_int32(v)) == 0, k
        got, _, _ = _prove_threads(backend, inp, world, cfg)
    finally:
        for k in switches:
            assert backend.L.cm_set_tuning(k.encode(), C.c_int32(DEFAULTS[k])) == 0
    for r in range(world):
        assert got[r].size == want.size and np.array_equal(got[r], want), (f"rank {r}", switches)
    inp.free()


def test_stream_ordered_ranks_all_opcode_segment(backend):
    from cairo_m_amd.lib import vm_run
    from cairo_m_amd.workloads import all_opcodes_program
    inp = vm_run(all_opcodes_program(24_000)[0], entry_pc=0, args=(), n_returns=0)
    p = backend.prove(inp)
    want = p.words().copy()
    p.free()
    got, _, _ = _prove_threads(backend, inp, 4)
    for r in range(4):
        assert got[r].size == want.size and np.array_equal(got[r], want), f"rank {r}"
    inp.free()
