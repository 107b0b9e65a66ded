"""The lists the device-side proof tail gathers by (cairo_m_amd/csrc/tail_device.hpp): U[k] / W[k] / F[k] of the sorted query set.
CPU test, no GPU: `cm_tail_list` (the host mirror the prover checks the device against, TailTables::build) is compared with a plain
restatement of Stwo's MerkleProver::decommit walk (the one MerkleTree::plan_decommit follows, merkle_tree.hpp) on trees of the three
shapes a proof has: a commitment tree (columns of several sizes, every size queried at the folds of one position set), an inner FRI
layer tree (columns at the leaves, whole sibling pairs opened) and the first FRI tree (columns of every size, sibling pairs opened at
every column-bearing layer).  The GPU tests then compare the kernels with this mirror through whole proofs."""
import ctypes as C
import random

import numpy as np

from cairo_m_amd.lib import load_library


def fold(positions, n=1):
    out = []
    for p in positions:
        if not out or out[-1] != p >> n:
            out.append(p >> n)
    return out


def pairs(positions):
    out = []
    for parent in fold(positions):
        out += [2 * parent, 2 * parent + 1]
    return out


def generic_walk(n_layers, col_layers, by_log):
    """MerkleProver::decommit: layers n_layers - 1 .. 0; `col_layers` = set of layers carrying columns; by_log[l] = sorted queried
    nodes of layer l.  Returns (hash witness [(layer, node)], column rows [(layer, node, is_query)]) in emission order."""
    hashes, rows = [], []
    last = []
    for layer in range(n_layers - 1, -1, -1):
        colq = by_log.get(layer, [])
        has_prev = layer + 1 < n_layers
        total, pi, qi = [], 0, 0
        while pi < len(last) or qi < len(colq):
            cands = []
            if pi < len(last):
                cands.append(last[pi] // 2)
            if qi < len(colq):
                cands.append(colq[qi])
            node = min(cands)
            if has_prev:
                if pi < len(last) and last[pi] == 2 * node:
                    pi += 1
                else:
                    hashes.append((layer + 1, 2 * node))
                if pi < len(last) and last[pi] == 2 * node + 1:
                    pi += 1
                else:
                    hashes.append((layer + 1, 2 * node + 1))
            isq = qi < len(colq) and colq[qi] == node
            if isq:
                qi += 1
            if layer in col_layers:
                rows.append((layer, node, isq))
            total.append(node)
        last = total
    return hashes, rows


def tail_list(L, S, log_domain, qmask, which, k):
    pos = np.ascontiguousarray(np.array(S, dtype=np.uint32))
    cap = 4 * len(S) + 8
    out = np.zeros(cap, dtype=np.uint32)
    n = C.c_uint32(0)
    rc = L.cm_tail_list(pos.ctypes.data_as(C.c_void_p), C.c_uint32(len(S)), C.c_uint32(log_domain), C.c_uint32(qmask), C.c_uint32(which),
                        C.c_uint32(k), out.ctypes.data_as(C.c_void_p), C.c_uint32(cap), C.byref(n))
    assert rc == 0
    assert n.value <= cap
    return [int(x) for x in out[:n.value]]


def _cases(rng, trials):
    for _ in range(trials):
        L0 = rng.randint(3, 16)
        nq = rng.randint(1, min(90, 1 << L0))
        if rng.random() < 0.3:   # clustered positions: neighbours and shared parents
            base = rng.randrange(1 << L0)
            S = sorted(set(min((1 << L0) - 1, base + rng.randint(0, 40)) for _ in range(nq)))
        else:
            S = sorted(set(rng.randrange(1 << L0) for _ in range(nq)))
        yield L0, S


def test_commitment_tree_walk_from_the_lists():
    L = load_library()
    rng = random.Random(11)
    for L0, S in _cases(rng, 120):
        top = rng.randint(1, L0)                                   # the tree's largest column: 2^top rows
        col_layers = set(l for l in range(0, top + 1) if rng.random() < 0.5) | {top}
        other_sizes = set(l for l in range(0, L0 + 1) if rng.random() < 0.5)   # sizes only OTHER trees have: queried all the same
        by_log = {l: fold(S, L0 - l) for l in col_layers | other_sizes | {L0}}
        hashes, rows = generic_walk(top + 1, col_layers, by_log)
        assert all(isq for _, _, isq in rows), "a commitment tree's column rows are all queried (no column witness)"
        got_h = [(j, x) for j in range(top, 0, -1) for x in tail_list(L, S, L0, 0, 1, L0 - j)]
        assert got_h == hashes, (L0, top, S)
        got_r = [(l, x) for l in range(top, -1, -1) if l in col_layers for x in tail_list(L, S, L0, 0, 0, L0 - l)]
        assert got_r == [(l, n) for l, n, _ in rows], (L0, top, S)


def test_inner_fri_layer_walk_from_the_lists():
    L = load_library()
    rng = random.Random(12)
    for L0, S in _cases(rng, 120):
        Li = rng.randint(1, L0)                                    # the layer: 2^Li values, queried at S folded L0 - Li times
        q = fold(S, L0 - Li)
        pos = pairs(q)
        witness = [p for p in pos if p not in set(q)]
        assert tail_list(L, S, L0, 0, 1, L0 - Li) == witness, (L0, Li, S)
        hashes, rows = generic_walk(Li + 1, {Li}, {Li: pos})
        assert all(isq for _, _, isq in rows)
        got = [(j, x) for j in range(Li - 1, 0, -1) for x in tail_list(L, S, L0, 0, 1, L0 - j)]
        assert got == hashes, (L0, Li, S)


def test_first_fri_tree_walk_from_the_lists():
    L = load_library()
    rng = random.Random(13)
    for L0, S in _cases(rng, 160):
        sizes = set(l for l in range(1, L0) if rng.random() < 0.6) | {L0}    # quotient column sizes; the largest spans the domain
        qmask = sum(1 << l for l in sizes)
        by_log = {l: pairs(fold(S, L0 - l)) for l in sizes}
        hashes, rows = generic_walk(L0 + 1, sizes, by_log)
        assert all(isq for _, _, isq in rows), "every node of a column-bearing layer is an opened position"
        got = [(j, x) for j in range(L0, 0, -1) for x in tail_list(L, S, L0, qmask, 2, L0 - j + 1)]
        assert got == hashes, (L0, sorted(sizes), S)
        for l in sizes:   # witness evaluations of the quotient columns of 2^l rows
            q = fold(S, L0 - l)
            assert tail_list(L, S, L0, qmask, 1, L0 - l) == [p for p in pairs(q) if p not in set(q)]


def test_tail_list_refuses_bad_arguments():
    """cm_tail_list checks what TailTables::build assumes: non-null pointers, positions strictly increasing and inside the domain."""
    L = load_library()
    out = np.zeros(64, dtype=np.uint32)
    n = C.c_uint32(0)

    def call(S, log_domain=6, out_p=out.ctypes.data_as(C.c_void_p), n_p=None, null_pos=False):
        pos = np.ascontiguousarray(np.array(S, dtype=np.uint32))
        return L.cm_tail_list(None if null_pos else pos.ctypes.data_as(C.c_void_p), C.c_uint32(len(S)), C.c_uint32(log_domain), C.c_uint32(0),
                              C.c_uint32(1), C.c_uint32(1), out_p, C.c_uint32(64), C.byref(n) if n_p is None else n_p)
    assert call([1, 5, 9]) == 0
    assert call([5, 1, 9]) != 0          # not sorted
    assert call([1, 5, 5]) != 0          # duplicate
    assert call([1, 5, 64]) != 0         # outside 2^6
    assert call([1, 5, 9], null_pos=True) != 0
    assert call([1, 5, 9], out_p=None) != 0
    assert call([1, 5, 9], n_p=C.c_void_p(0)) != 0
