"""The shortcut FriPhase::plan_decommit takes (cairo_m_amd/csrc/prover_common.hpp): the decommitment walks of ALL inner FRI layer
trees come from one family of lists, W[m] = the siblings — missing from the set — of the query positions folded m times.  This is
a CPU restatement of both sides: the generic walk (Stwo MerkleProver::decommit as MerkleTree::plan_decommit restates it, for a tree
whose only columns sit at the leaves) and the list form; they must name the same witness nodes in the same order for every layer.
The C++ code itself is covered by the GPU parity tests (whole proofs against the oracle, whose verifier walks generically)."""
import random


def fold(positions):
    out = []
    for p in positions:
        if not out or out[-1] != p >> 1:
            out.append(p >> 1)
    return out


def decommit_positions(queries):
    """compute_decommitment_positions_and_witness_evals: whole sibling pairs; witness = the pair members that are not queried"""
    pos, wit = [], []
    qs = set(queries)
    for parent in fold(queries):
        for p in (2 * parent, 2 * parent + 1):
            pos.append(p)
            if p not in qs:
                wit.append(p)
    return pos, wit


def generic_walk(log, leaf_positions):
    """hash witness of a tree with 2^log leaves queried at `leaf_positions` (sorted): [(level, node index)] in emission order"""
    out = []
    last = list(leaf_positions)
    for layer in range(log - 1, -1, -1):   # parents at `layer`, children at layer + 1
        total, pi = [], 0
        while pi < len(last):
            node = last[pi] >> 1
            if last[pi] == 2 * node:
                pi += 1
            else:
                out.append((layer + 1, 2 * node))
            if pi < len(last) and last[pi] == 2 * node + 1:
                pi += 1
            else:
                out.append((layer + 1, 2 * node + 1))
            total.append(node)
        last = total
    return out


def sibling_lists(positions, log):
    W, B = [], list(positions)
    while True:
        s = set(B)
        W.append([b ^ 1 for b in B if (b ^ 1) not in s])
        if log == 0:
            break
        B, log = fold(B), log - 1
    return W


def test_all_layers_from_one_family_of_sibling_lists():
    rng = random.Random(7)
    for trial in range(200):
        log0 = rng.randint(2, 14)
        n_q = rng.randint(1, min(70, 1 << log0))
        q0 = sorted(set(rng.randrange(1 << log0) for _ in range(n_q)))
        W = sibling_lists(q0, log0)
        q, n_layers = q0, rng.randint(1, log0 - 1)
        for i in range(n_layers):
            log_i = log0 - i
            pos, wit = decommit_positions(q)
            assert wit == W[i], (trial, i)
            want = generic_walk(log_i, pos)
            got = [(log_i - k, idx) for k in range(1, log_i) for idx in W[i + k]]
            assert got == want, (trial, i, log_i, q)
            q = fold(q)
