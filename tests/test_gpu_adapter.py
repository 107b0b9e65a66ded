"""Device-side adapter (cm_adapt_segment_device) vs the host adapter (cm_vm_run / cm_synth_fibonacci,
the sequential restatement of adapter/mod.rs:97-193 that tests/test_adapter.py pins with the reference's
Memory::push known-answer tests): every array of the ProverInput must be identical, element by element, and the
proof built from the device-adapted input must be bit-identical to the proof of the host-adapted one."""
import numpy as np
import pytest

from cairo_m_amd.lib import (prover_input_arrays, synth_fibonacci, synth_fibonacci_segment, vm_run, vm_segment)

pytestmark = pytest.mark.gpu


def _same(a, b):
    assert a.keys() == b.keys()
    for k in a:
        if isinstance(a[k], list):
            assert a[k] == b[k], k
        else:
            assert a[k].shape == b[k].shape, (k, a[k].shape, b[k].shape)
            assert np.array_equal(a[k], b[k]), (k, np.nonzero(a[k] != b[k])[0][:5])


def _check(backend, host_input, host_segment, prove=True):
    dev = backend.adapt_segment(host_segment)
    back = backend.download_input(dev)
    _same(prover_input_arrays(host_input.view), prover_input_arrays(back.view))
    if prove:
        p1 = backend.prove_device(dev)
        p2 = backend.prove(host_input)
        assert np.array_equal(p1.words(), p2.words())
        p1.free(); p2.free()
    back.free()
    backend.free_input(dev)


@pytest.mark.parametrize("n", [3, 1000])
def test_fibonacci_segment(backend, n):
    hi, hs = synth_fibonacci(n), synth_fibonacci_segment(n)
    _check(backend, hi, hs)
    hi.free(); hs.free()


def test_all_opcode_programs(backend):
    from tests.test_oracle_air import felt_program, u32_program, u32_loop_program
    for prog, nret in ((felt_program(), 1), (u32_program(), 0), (u32_loop_program(50), 0)):
        hi = vm_run(prog, entry_pc=0, args=(), n_returns=nret)
        hs = vm_segment(prog, entry_pc=0, args=(), n_returns=nret)
        _check(backend, hi, hs)
        hi.free(); hs.free()


def test_continuation_segments(backend):
    """segments cut every max_steps: the device adapter starts from the memory at segment start"""
    from tests.test_oracle_air import CHAIN_PROG
    for s in range(4):
        hi = vm_run(CHAIN_PROG, max_steps=2, segment=s)
        hs = vm_segment(CHAIN_PROG, max_steps=2, segment=s)
        _check(backend, hi, hs, prove=False)
        hi.free(); hs.free()


def test_metric_config_with_clock_updates(backend):
    """fibonacci_loop n = 419 000 (4.19M steps): cells idle for more than 2^20 - 1 steps produce clock-update rows."""
    hi, hs = synth_fibonacci(419_000), synth_fibonacci_segment(419_000)
    a = prover_input_arrays(hi.view)
    assert a["clock_updates"].shape[0] > 0
    _check(backend, hi, hs)
    hi.free(); hs.free()


def test_partial_merkle_trees_on_device(backend, monkeypatch):
    """CM_ADAPTER_DEVICE_TREE_MIN=1 forces the GPU partial-Merkle-tree builder (Poseidon2, level by level; normally
    used from 2048 boundary cells up): node lists, roots and proofs must still equal the host adapter's."""
    from tests.test_oracle_air import felt_program, u32_loop_program
    monkeypatch.setenv("CM_ADAPTER_DEVICE_TREE_MIN", "1")
    hi, hs = synth_fibonacci(60), synth_fibonacci_segment(60)
    _check(backend, hi, hs)
    hi.free(); hs.free()
    for prog, nret in ((felt_program(), 1), (u32_loop_program(20), 0)):
        hi = vm_run(prog, entry_pc=0, args=(), n_returns=nret)
        hs = vm_segment(prog, entry_pc=0, args=(), n_returns=nret)
        _check(backend, hi, hs)
        hi.free(); hs.free()


def scatter_store_program(n, base=100):
    """A loop that writes n distinct memory cells through a double dereference ([[fp+8] + [fp+11]] = [fp+12]):
    the boundary memory grows to n cells, which takes the device adapter's GPU Merkle-tree path (>= 2048 cells)."""
    P = 2**31 - 1
    return [
        [9, 0, 20],            # pc 0: i = 0
        [43, 0, 8],            # pc 1: [fp+8] = fp
        [9, 7, 12],            # pc 2: value = 7
        [9, n, 30],            # pc 3: counter = n
        [4, 20, base, 11],     # pc 4: off = i + base                     <- loop head
        [45, 8, 11, 12],       # pc 5: [[fp+8] + off] = value  (cell fp + base + i)
        [4, 20, 1, 21],        # pc 6: i' = i + 1
        [4, 21, 0, 20],        # pc 7: i = i'
        [4, 30, P - 1, 31],    # pc 8: c' = c - 1
        [4, 31, 0, 30],        # pc 9: c = c'
        [14, 30, P - 6],       # pc 10: jnz c -> pc 4
        [11],                  # pc 11: ret
    ]


def test_large_boundary_memory_uses_device_trees(backend):
    prog = scatter_store_program(3000)
    hi = vm_run(prog, entry_pc=0, args=(), n_returns=0)
    hs = vm_segment(prog, entry_pc=0, args=(), n_returns=0)
    a = prover_input_arrays(hi.view)
    assert a["initial_memory"].shape[0] >= 3000 and a["initial_tree"].shape[0] > 10_000
    _check(backend, hi, hs)
    hi.free(); hs.free()


def test_wide_addresses(backend):
    """The address sort only runs over the bits the largest address of the log uses: cells near the top of the 28-bit address space
    (adapter/merkle.rs tree height 30 = address bits + 2) next to the program's own small addresses."""
    prog = scatter_store_program(300, base=(1 << 27) + 12345)
    hi = vm_run(prog, entry_pc=0, args=(), n_returns=0)
    hs = vm_segment(prog, entry_pc=0, args=(), n_returns=0)
    a = prover_input_arrays(hi.view)
    assert int(a["data_accesses"][:, 0].max()) >= 1 << 27
    _check(backend, hi, hs)
    hi.free(); hs.free()


def test_heap_segments(backend, oracle):
    """The compiler's `new felt[3]` program (tests/casm_fixtures.py heap_program): cells at the top of the 2^28-cell address space.
    Whole: device adapter == host adapter, proof == the oracle's.  Cut every 8 steps: segments 2 and 3 start with a heap
    (cm_runner_segment.initial_heap, ABI revision 6) whose cells the device kernels must find for previous values and which belong
    to the boundary memory whether the segment touches them or not."""
    from tests.casm_fixtures import heap_program
    prog, entry, nret, _ = heap_program()
    hi = vm_run(prog, entry_pc=entry, n_returns=nret)
    hs = vm_segment(prog, entry_pc=entry, n_returns=nret)
    _check(backend, hi, hs)
    p = backend.prove(hi)
    want, _ = oracle.prove(hi.view)
    assert np.array_equal(p.words(), want)
    p.free(); hi.free(); hs.free()
    from cairo_m_amd.lib import runner_segment_arrays
    n_heap = []
    for s in range(4):
        hi = vm_run(prog, entry_pc=entry, n_returns=nret, max_steps=8, segment=s)
        hs = vm_segment(prog, entry_pc=entry, n_returns=nret, max_steps=8, segment=s)
        n_heap.append(runner_segment_arrays(hs.view)["initial_heap"].shape[0])
        _check(backend, hi, hs)
        hi.free(); hs.free()
    assert n_heap[0] == 0 and n_heap[3] == 3, n_heap


def test_runner_artifact_files_to_proof(backend, oracle, tmp_path):
    """SURVEY 8 f-3 end to end: the runner's on-disk artifacts — binary trace ((fp, pc) LE u32 pairs, execution.rs:28-66),
    binary memory trace (u32 program_length header + (address, 4 value words) records, io.rs:38-80) and the compiled
    Program JSON (program.rs:143-170) — are written to files, read back, fed through cm_segment_from_artifacts ->
    cm_adapt_segment_device -> cm_prove_device; the proof equals the oracle's proof of the in-memory host-adapted input."""
    import ctypes as C
    from cairo_m_amd.lib import HostSegment, load_program_json, program_to_json, vm_run, vm_segment
    from tests.test_oracle_air import felt_program

    class Seg(C.Structure):
        _fields_ = [("trace", C.c_void_p), ("n_trace", C.c_uint64), ("memory_trace", C.c_void_p), ("n_memory_trace", C.c_uint64),
                    ("initial_memory", C.c_void_p), ("n_initial_memory", C.c_uint64), ("ranges", C.c_uint32 * 6)]

    L = backend.L
    (tmp_path / "program.json").write_text(program_to_json(felt_program(), {"main": {"pc": 0, "returns": [{"name": "r", "ty": "Felt"}]}}))
    cells, entry = load_program_json((tmp_path / "program.json").read_text())
    main = entry["main"]
    hs = vm_segment(cells, entry_pc=main["pc"], args=(), n_returns=main["n_returns"])
    n = C.c_uint64(0)
    L.cm_segment_serialize_trace(hs.view, None, C.c_uint64(0), C.byref(n))
    tb = (C.c_uint8 * n.value)()
    assert L.cm_segment_serialize_trace(hs.view, tb, C.c_uint64(n.value), C.byref(n)) == 0
    L.cm_segment_serialize_memory_trace(hs.view, 1, None, C.c_uint64(0), C.byref(n))
    mb = (C.c_uint8 * n.value)()
    assert L.cm_segment_serialize_memory_trace(hs.view, 1, mb, C.c_uint64(n.value), C.byref(n)) == 0
    (tmp_path / "trace.bin").write_bytes(bytes(tb))
    (tmp_path / "memory.bin").write_bytes(bytes(mb))
    s = C.cast(hs.view, C.POINTER(Seg)).contents
    init = np.ctypeslib.as_array(C.cast(s.initial_memory, C.POINTER(C.c_uint32)), shape=(int(4 * s.n_initial_memory),)).copy()
    ranges = (C.c_uint32 * 6)(*list(s.ranges))
    hs.free()
    # ---- a fresh reader: only the files (+ the initial memory / public ranges the reference keeps out of band)
    t = (tmp_path / "trace.bin").read_bytes()
    m = (tmp_path / "memory.bin").read_bytes()
    h2 = C.c_void_p()
    assert L.cm_segment_from_artifacts(t, C.c_uint64(len(t)), m, C.c_uint64(len(m)), 1, init.ctypes.data_as(C.POINTER(C.c_uint32)),
                                       C.c_uint64(init.size // 4), ranges, C.byref(h2)) == 0
    seg = HostSegment(L, h2)
    dev = backend.adapt_segment(seg)
    p = backend.prove_device(dev)
    hi = vm_run(felt_program(), entry_pc=0, args=(), n_returns=1)
    want, _ = oracle.prove(hi.view)
    assert np.array_equal(p.words(), want)
    assert p.verify()[0] == 0
    p.free()
    backend.free_input(dev)
    seg.free(); hi.free()
