"""`write_interaction_trace` as the GPU runs it: for each of the 34 components (the 26 opcode components, memory, merkle, clock_update, poseidon2, the four lookup tables) the reference-derived TRACE cells
(tests/golden/air_witness_vectors.npz, from the reference's `write_trace` closures) go through the HIP k_logup / LogUp-tail kernels
(cm_interaction_write) under fixed relation parameters, and the interaction columns must equal the reference-derived LogUp
vectors (tests/golden/air_logup_vectors.npz: tools/rsref/rs_logup.py interprets the reference's `write_interaction_trace` text —
which lookup tuples pair up in which column, every numerator and denominator — on the lookup data the same run's `write_trace`
produced).  Nothing of cairo_m_amd/csrc/air/*.hpp, of the oracle or of the library's field code is on the expected side; the
arithmetic HERE is QM31 addition only.

Column j < last: the HIP column (4 coordinate columns) == sum of the fractions of columns 0..j of the row — exactly what
LogupTraceGenerator::finalize_col leaves there.  The LAST column is prefix-summed over the rows by finalize_last; with T(r) the
row's total and shift = claimed_sum / n it satisfies c(r) - c(prev(r)) = T(r) - shift for the cyclic predecessor prev(r) of the
trace domain, so — independent of the row order convention — the multiset { c(r) - T(r) + shift } equals the multiset { c(r) },
and the returned claimed sum equals sum_r T(r)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WIT = np.load(os.path.join(ROOT, "tests", "golden", "air_witness_vectors.npz"))
LOG = np.load(os.path.join(ROOT, "tests", "golden", "air_logup_vectors.npz"))
from tests.test_air_witness_golden import OPCODE_FILES  # noqa: E402

P = 2**31 - 1
N_REL, MAX_REL, N_PP = 8, 16, 7


def cmul(x, y):
    return ((x[0] * y[0] - x[1] * y[1]) % P, (x[0] * y[1] + x[1] * y[0]) % P)


def qmul(x, y):
    a, b, c, d = x[:2], x[2:], y[:2], y[2:]
    ac, bd = cmul(a, c), cmul(b, d)
    r = cmul(bd, (2, 1))
    ad, bc = cmul(a, d), cmul(b, c)
    return ((ac[0] + r[0]) % P, (ac[1] + r[1]) % P, (ad[0] + bc[0]) % P, (ad[1] + bc[1]) % P)


def relation_words():
    """cm_relations: z[8][4] then alpha_pow[8][16][4]"""
    z = LOG["rel_z"].astype(np.uint32)
    pw = np.zeros((N_REL, MAX_REL, 4), dtype=np.uint32)
    for r in range(N_REL):
        a, cur = tuple(int(x) for x in LOG["rel_alpha"][r]), (1, 0, 0, 0)
        for i in range(MAX_REL):
            pw[r, i] = cur
            cur = qmul(cur, a)
    return np.concatenate([z.reshape(-1), pw.reshape(-1)])


def test_same_run_as_the_witness_vectors():
    assert int(LOG["iters"][0]) == int(WIT["iters"][0]) and int(LOG["seed"][0]) == int(WIT["seed"][0])
    assert all(f in LOG.files for f in NAMES)


# component id -> name in the golden files: the 26 opcode components, then memory / merkle / clock_update (the builtins whose
# write_trace closure has the regular shape; clock_update on the synthetic entries of tools/rsref/rs_witness.py)
NAMES = list(OPCODE_FILES) + ["memory", "merkle", "clock_update", "poseidon2", "range_check_8", "range_check_16", "range_check_20", "bitwise"]
# the four lookup tables (ids 30..33): trace = the multiplicity column, the table entries come in through the preprocessed
# columns (air::PreprocId: bitwise 0..3, range_check_8 / 16 / 20 = 4 / 5 / 6); 64 seeded rows stored with the vectors
TABLE_PP = {"range_check_8": [4], "range_check_16": [5], "range_check_20": [6], "bitwise": [0, 1, 2, 3]}


@pytest.mark.parametrize("cid", range(34), ids=NAMES)
def test_hip_logup_columns_equal_reference_derived_fractions(backend, cid):
    name = NAMES[cid]
    want = LOG[name].astype(np.int64)                               # (n_cols, n, 4)
    trace = LOG[name + "_mults"][None, :] if name in TABLE_PP else WIT[name]     # (n_trace, n)
    n = trace.shape[1]
    log = n.bit_length() - 1
    n_trace, n_inter, _ = backend.component_info(cid)
    assert trace.shape[0] == n_trace and want.shape == (n_inter // 4, n, 4)
    h_tr = [backend.upload(np.ascontiguousarray(trace[c])) for c in range(n_trace)]
    pp = np.zeros((N_PP, n), dtype=np.uint32)                       # only the lookup tables read preprocessed columns
    for k, idx in enumerate(TABLE_PP.get(name, [])):
        pp[idx] = LOG[name + "_values"][k]
    h_pp = [backend.upload(np.ascontiguousarray(pp[i])) for i in range(N_PP)]
    h_out = [backend.col_alloc(n) for _ in range(n_inter)]
    try:
        cs = backend.interaction_write(cid, h_tr, h_pp, log, relation_words(), h_out)
        got = np.stack([backend.download(h, n) for h in h_out]).astype(np.int64).reshape(n_inter // 4, 4, n)
        last = n_inter // 4 - 1
        for j in range(last):
            bad = np.argwhere(got[j].T != want[j])
            assert bad.size == 0, f"{name}: LogUp column {j}: first differing (row, coordinate) {bad[:4].tolist()}"
        total = want[last]                                              # T(r): all fractions of row r
        claimed = total.sum(axis=0) % P
        assert [int(x) for x in cs] == [int(x) for x in claimed], f"{name}: claimed sum"
        shift = claimed * pow(n, P - 2, P) % P
        c = got[last].T                                                 # (n, 4)
        lhs = (c - total + shift) % P
        key = lambda a: sorted(map(tuple, a.tolist()))
        assert key(lhs) == key(c), f"{name}: last column is not the running sum of the row totals minus the shift"
        assert np.any(total % P)                                        # the vectors are not trivially zero
    finally:
        for h in h_tr + h_pp + h_out:
            backend.col_free(h)
