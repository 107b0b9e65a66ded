"""Adapter known answers restated from the reference's own unit tests
(/root/reference/crates/prover/src/adapter/memory.rs:545-858: Memory::push) through the C-ABI test hook.
CPU only (host code of the library)."""
import ctypes as C

import numpy as np
import pytest

from cairo_m_amd.lib import load_library

P = 2**31 - 1
RC20_LIMIT = (1 << 20) - 1


def run(script, preload=(), queries=()):
    L = load_library()
    pre = np.array([w for e in preload for w in e], dtype=np.uint32)
    scr = np.array([w for e in script for w in e], dtype=np.uint32)
    res = np.zeros(5 * len(script), dtype=np.uint32)
    ncu = C.c_uint32(0)
    cu = np.zeros(6 * 64, dtype=np.uint32)
    q = np.array(list(queries), dtype=np.uint32)
    st = np.zeros(14 * max(1, len(queries)), dtype=np.uint32)
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint32))
    rc = L.cm_adapter_memory_script(p(pre), C.c_uint32(len(preload)), p(scr), C.c_uint32(len(script)), p(res), C.byref(ncu),
                                    p(cu), C.c_uint32(64), p(q), C.c_uint32(len(queries)), p(st))
    assert rc == 0
    return res.reshape(-1, 5), cu[: 6 * ncu.value].reshape(-1, 6), st.reshape(-1, 2, 7)


def test_memory_push_first_entry():  # memory.rs:545-584
    res, cu, st = run([(100, 1, 2, 3, 4, 10)], queries=[100])
    assert list(res[0]) == [0, 1, 2, 3, 4]           # prev_clock 0, prev_val = value
    assert list(st[0][1]) == [1, 1, 2, 3, 4, 10, P - 1]   # final: (value, clock 10, -1)
    assert list(st[0][0]) == [1, 1, 2, 3, 4, 0, 1]        # initial: (value, 0, 1)
    assert len(cu) == 0


def test_memory_push_same_address():  # memory.rs:586-634
    res, _, st = run([(100, 1, 2, 3, 4, 10), (100, 5, 6, 7, 8, 20)], queries=[100])
    assert list(res[1]) == [10, 1, 2, 3, 4]
    assert list(st[0][1]) == [1, 5, 6, 7, 8, 20, P - 1]
    assert list(st[0][0]) == [1, 1, 2, 3, 4, 0, 1]


def test_memory_push_different_addresses():  # memory.rs:636-698
    res, _, st = run([(100, 1, 2, 3, 4, 10), (200, 9, 10, 11, 12, 30)], queries=[100, 200])
    assert list(res[1]) == [0, 9, 10, 11, 12]
    assert list(st[0][1]) == [1, 1, 2, 3, 4, 10, P - 1] and list(st[1][1]) == [1, 9, 10, 11, 12, 30, P - 1]
    assert list(st[0][0]) == [1, 1, 2, 3, 4, 0, 1] and list(st[1][0]) == [1, 9, 10, 11, 12, 0, 1]


def test_memory_push_multiple_large_clock_deltas():  # memory.rs:700-737
    delta = 3 * RC20_LIMIT + 500
    res, cu, _ = run([(100, 1, 2, 3, 4, 10), (100, 5, 6, 7, 8, 10 + delta)])
    assert len(cu) == 3
    assert [int(r[1]) for r in cu] == [10, 10 + RC20_LIMIT, 10 + 2 * RC20_LIMIT]
    assert int(res[1][0]) == 10 + 3 * RC20_LIMIT  # prev_clock = last inserted step


def test_memory_push_no_clock_update_for_small_delta():  # memory.rs:739-760
    _, cu, _ = run([(100, 1, 2, 3, 4, 10), (100, 5, 6, 7, 8, 10 + RC20_LIMIT - 1)])
    assert len(cu) == 0


def test_memory_push_with_preloaded_memory():  # memory.rs:762-858
    pre = [(0, 10, 20, 30, 40), (1, 50, 60, 70, 80)]
    res, _, st = run([(0, 10, 20, 30, 40, 5)], preload=pre, queries=[0, 1])
    assert list(res[0]) == [0, 10, 20, 30, 40]
    assert list(st[0][0]) == [1, 10, 20, 30, 40, 0, 1]          # initial multiplicity flips to 1
    assert list(st[0][1]) == [1, 10, 20, 30, 40, 5, P - 1]
    assert list(st[1][0]) == [1, 50, 60, 70, 80, 0, 0]
    res, _, st = run([(0, 10, 20, 30, 40, 5), (0, 100, 200, 300, 400, 10)], preload=pre, queries=[0])
    assert list(res[1]) == [5, 10, 20, 30, 40]
    assert list(st[0][1]) == [1, 100, 200, 300, 400, 10, P - 1]


def test_poseidon2_host_matches_reference_kat():
    import json, os
    kat = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "poseidon2_kat.json")))
    L = load_library()
    s = np.array(kat["input"], dtype=np.uint32)
    L.cm_poseidon2_permute(s.ctypes.data_as(C.POINTER(C.c_uint32)))
    assert [f"{int(x):08x}" for x in s] == kat["output_hex"]


def test_runner_segment_view_matches_host_input():
    """cm_vm_segment / cm_synth_fibonacci_segment expose the raw runner output (the device adapter's input):
    trace = steps + 1 states, the memory log holds one entry per instruction word cell and operand access."""
    import ctypes as C
    from cairo_m_amd.lib import prover_input_arrays, synth_fibonacci, synth_fibonacci_segment

    class Seg(C.Structure):
        _fields_ = [("trace", C.c_void_p), ("n_trace", C.c_uint64), ("memory_trace", C.c_void_p), ("n_memory_trace", C.c_uint64),
                    ("initial_memory", C.c_void_p), ("n_initial_memory", C.c_uint64), ("ranges", C.c_uint32 * 6)]

    hi, hs = synth_fibonacci(25), synth_fibonacci_segment(25)
    seg = C.cast(hs.view, C.POINTER(Seg)).contents
    a = prover_input_arrays(hi.view)
    assert seg.n_trace == hi.steps + 1 == 10 * 25 + 12 + 1
    n_acc = a["data_accesses"].shape[0]
    # fibonacci uses one-word instructions only: entries = steps (fetches) + operand accesses
    assert seg.n_memory_trace == hi.steps + n_acc
    assert list(seg.ranges) == a["ranges"]
    tr = np.ctypeslib.as_array(C.cast(seg.trace, C.POINTER(C.c_uint32)), shape=(int(seg.n_trace), 2))
    assert [int(tr[0][0]), int(tr[0][1]), int(tr[-1][0]), int(tr[-1][1])] == a["regs"]
    hi.free(); hs.free()


def test_runner_artifact_wire_formats():
    """Byte layout of the runner artifacts (execution.rs:28-66: `fp` before `pc`; address + 4 value words; optional
    u32 program_length header, io.rs:76-80) and the round trip segment -> files -> segment."""
    import ctypes as C
    from cairo_m_amd.lib import load_library, synth_fibonacci_segment, HostSegment
    L = load_library()

    class Seg(C.Structure):
        _fields_ = [("trace", C.c_void_p), ("n_trace", C.c_uint64), ("memory_trace", C.c_void_p), ("n_memory_trace", C.c_uint64),
                    ("initial_memory", C.c_void_p), ("n_initial_memory", C.c_uint64), ("ranges", C.c_uint32 * 6)]

    def arrays(view):
        s = C.cast(view, C.POINTER(Seg)).contents
        f = lambda p, n: np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(int(n),)).copy()
        return f(s.trace, 2 * s.n_trace), f(s.memory_trace, 5 * s.n_memory_trace), f(s.initial_memory, 4 * s.n_initial_memory), list(s.ranges)

    hs = synth_fibonacci_segment(7)
    tr, mem, init, ranges = arrays(hs.view)
    n = C.c_uint64(0)
    assert L.cm_segment_serialize_trace(hs.view, None, C.c_uint64(0), C.byref(n)) == 0 and n.value == 4 * tr.size
    tb = (C.c_uint8 * n.value)()
    assert L.cm_segment_serialize_trace(hs.view, tb, C.c_uint64(n.value), C.byref(n)) == 0
    words = np.frombuffer(bytes(tb), dtype="<u4").reshape(-1, 2)
    assert np.array_equal(words[:, 0], tr.reshape(-1, 2)[:, 1])      # fp first
    assert np.array_equal(words[:, 1], tr.reshape(-1, 2)[:, 0])      # then pc
    assert L.cm_segment_serialize_memory_trace(hs.view, 1, None, C.c_uint64(0), C.byref(n)) == 0 and n.value == 4 + 4 * mem.size
    mb = (C.c_uint8 * n.value)()
    assert L.cm_segment_serialize_memory_trace(hs.view, 1, mb, C.c_uint64(n.value), C.byref(n)) == 0
    mw = np.frombuffer(bytes(mb), dtype="<u4")
    assert mw[0] == ranges[1] - ranges[0] and np.array_equal(mw[1:], mem)
    h2 = C.c_void_p()
    r = (C.c_uint32 * 6)(*ranges)
    assert L.cm_segment_from_artifacts(tb, C.c_uint64(len(tb)), mb, C.c_uint64(len(mb)), 1, init.ctypes.data_as(C.POINTER(C.c_uint32)),
                                       C.c_uint64(init.size // 4), r, C.byref(h2)) == 0
    back = HostSegment(L, h2)
    tr2, mem2, init2, ranges2 = arrays(back.view)
    assert np.array_equal(tr, tr2) and np.array_equal(mem, mem2) and np.array_equal(init, init2) and ranges == ranges2
    # a truncated file is rejected
    assert L.cm_segment_from_artifacts(tb, C.c_uint64(len(tb) - 3), mb, C.c_uint64(len(mb)), 1, init.ctypes.data_as(C.POINTER(C.c_uint32)),
                                       C.c_uint64(init.size // 4), r, C.byref(h2)) != 0
    back.free(); hs.free()


def test_compiled_program_json_roundtrip():
    """Program JSON in the serde shape of crates/common/src/program.rs:143-170 (instructions as hex-string arrays,
    instruction.rs:609-655): parse, run through the VM + adapter, same ProverInput as the word-list form."""
    from cairo_m_amd.lib import load_program_json, program_to_json, prover_input_arrays, vm_run
    from tests.test_oracle_air import felt_program
    prog = felt_program()
    text = program_to_json(prog, {"main": {"pc": 0, "returns": [{"name": "r", "ty": "Felt"}]},
                                  "f": {"pc": 26, "params": [{"name": "a", "ty": "U32"}, {"name": "p", "ty": {"Pointer": {"element": "Felt", "len": None}}}]}})
    assert '"0x9"' in text and '"Instruction"' in text
    cells, entry = load_program_json(text)
    assert cells == prog
    assert entry["main"] == {"pc": 0, "n_params": 0, "n_returns": 1} and entry["f"]["n_params"] == 3
    a = prover_input_arrays(vm_run(cells, entry_pc=entry["main"]["pc"], args=(), n_returns=entry["main"]["n_returns"]).view)
    b = prover_input_arrays(vm_run(prog, entry_pc=0, args=(), n_returns=1).view)
    assert all(np.array_equal(a[k], b[k]) if hasattr(a[k], "shape") else a[k] == b[k] for k in a)
    with pytest.raises(ValueError):
        load_program_json('{"data": [], "entrypoints": {}, "metadata": {}, "extra": 1}')


def _partial_tree(cells, initial=True, ranges=(0, 0, 0, 0, 0, 0)):
    import ctypes as C
    from cairo_m_amd.lib import load_library
    L = load_library()
    c = np.ascontiguousarray(np.array(cells, dtype=np.uint32).reshape(-1, 5))
    cap = 4096
    out = np.zeros((cap, 8), dtype=np.uint32)
    n, root = C.c_uint64(0), C.c_uint32(0)
    u = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint32))
    assert L.cm_adapter_partial_tree(u(c), C.c_uint32(c.shape[0]), C.c_int32(1 if initial else 0), (C.c_uint32 * 6)(*ranges), u(out),
                                     C.c_uint64(cap), C.byref(n), C.byref(root)) == 0
    return out[:n.value], root.value


def test_partial_merkle_tree_shapes(oracle):
    """The reference's build_partial_merkle_tree tests (adapter/merkle.rs:302-423), restated on the host builder: empty
    memory -> empty tree; one cell -> nodes up to the root; two cells -> the depth-30 leaf pairs hold the QM31 words
    (address a -> leaves 4a..4a+3); addresses 0 and 2^28 - 1 -> nodes at every depth 30..1.  Plus: the root equals the
    hash chain recomputed with the oracle's Poseidon2 (pinned by the reference KAT)."""
    nodes, root = _partial_tree([])
    assert nodes.shape[0] == 0 and root == 0                                  # test_empty_tree
    nodes, root = _partial_tree([[5, 42, 0, 0, 0]])                           # test_single_element_tree
    assert nodes.shape[0] > 0 and root != 0
    assert sorted(set(nodes[:, 1].tolist())) == list(range(1, 31))            # one path: every depth 30..1
    nodes, root = _partial_tree([[0, 10, 11, 12, 13], [1, 20, 21, 22, 23]])   # test_multiple_elements_tree
    find = lambda idx, depth: nodes[(nodes[:, 0] == idx) & (nodes[:, 1] == depth)][0]
    assert find(0, 30)[2:4].tolist() == [10, 11] and find(2, 30)[2:4].tolist() == [12, 13] and find(4, 30)[2:4].tolist() == [20, 21]
    nodes, root = _partial_tree([[0, 1, 0, 0, 0], [(1 << 28) - 1, 2, 0, 0, 0]])  # test_tree_builds_to_root
    assert nodes[:, 1].min() == 1 and nodes[:, 1].max() == 30
    # every node's parent_value = poseidon2(left, right)[0], and the depth-1 node's parent value is the root
    for nd in nodes[::7]:
        st = np.zeros(16, dtype=np.uint32)
        st[0], st[1] = nd[2], nd[3]
        assert oracle.poseidon2_permute(st)[0] == nd[4]
    assert nodes[nodes[:, 1] == 1][0][4] == root
    # multiplicities: leaves of a public (program-range) cell count twice in the initial tree (adapter/merkle.rs:210-223)
    nodes, _ = _partial_tree([[3, 7, 0, 0, 0]], initial=True, ranges=(0, 8, 0, 0, 0, 0))
    leaf = nodes[(nodes[:, 1] == 30) & (nodes[:, 0] == 12)][0]
    assert leaf[5] == 2 and leaf[6] == 2


def test_bundle_order_inside_a_component_follows_the_reference():
    """`states_by_opcodes` is one Vec per OPCODE in step order (adapter/mod.rs:118-130) and Claim::write_trace concatenates
    the variants of a component in `define_opcodes!` order (components/opcodes/mod.rs:51-58, 223-268): the rows of
    store_fp_fp are all StoreAddFpFp steps, then all StoreSubFpFp, StoreMulFpFp, StoreDivFpFp — NOT interleaved in step
    order.  Same for jmp_imm (abs, rel) and store_fp_imm (add, mul)."""
    from cairo_m_amd.lib import prover_input_arrays, vm_run
    prog = [
        [9, 7, 0], [9, 3, 1],
        [3, 0, 1, 2],        # div first
        [0, 0, 1, 3],        # add
        [2, 0, 1, 4],        # mul
        [1, 0, 1, 5],        # sub
        [0, 2, 3, 6],        # add again
        [6, 6, 5, 7],        # mul imm
        [4, 7, 9, 8],        # add imm
        [13, 2],             # jmp rel +2
        [9, 1, 9],           # skipped
        [12, 12],            # jmp abs 12
        [11],
    ]
    a = prover_input_arrays(vm_run(prog, entry_pc=0, args=(), n_returns=0).view)
    fpfp = a["bundles6"]
    assert fpfp[:, 4].tolist() == [0, 0, 1, 2, 3]                 # opcode variants grouped in macro order ...
    assert fpfp[:2, 2].tolist() == sorted(fpfp[:2, 2].tolist())   # ... step order (clock) kept inside a variant
    assert fpfp[:, 2].tolist() == [4, 7, 6, 5, 3]
    assert a["bundles7"][:, 4].tolist() == [4, 6] and a["bundles2"][:, 4].tolist() == [12, 13]
