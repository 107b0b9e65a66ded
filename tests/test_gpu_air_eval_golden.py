"""The `eval` half of the AIR descriptions as the GPU runs it: the reference-derived vectors of tests/golden/air_eval_vectors.json
(made mechanically from the reference's `FrameworkEval::evaluate` text by tools/rsref/rs_eval.py) go through the HIP
`DomainEval` / `k_constraints<C>` kernels (cm_constraints_accumulate) on a 2^4-row component, i.e. a 2^5-row evaluation domain.

What is checked against WHAT: the rows of arbitrary field elements are placed on the evaluation domain as the component's
trace columns, the interaction columns are zero, relation challenges, constraint coefficients and the claimed sum are random.
Then every LogUp constraint (c_j - c_{j-1}) * D_j - N_j collapses to -N_j — the last one, which carries the cumulative-sum shift,
to shift * D_last - N_last — and the kernel's accumulator must equal

    ( sum_k coeff_k * C_k  -  sum_j coeff_{n_base + j} * N_j  +  coeff_last * shift * D_last ) / vanishing(row)

where C_k are the golden constraint values and N_j the numerators of the paired fractions built from the golden relation entries
(multiplicity, tuple) — computed HERE in plain Python integers (M31 / QM31 arithmetic below), from the vectors alone.  Nothing of
cairo_m_amd/csrc/air/*.hpp, of the oracle or of the library's field code is on the expected side, so this pins the GPU
instantiation of all 34 components (column order, every constraint, every relation entry, the pairing, the
coefficient order, the vanishing-polynomial inverse) directly to the reference-derived data.  (tests/test_air_eval_golden.py pins
the CPU instantiation of the same descriptions.)"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "air_eval_vectors.json")))["components"]
REL_ID = {"registers": 0, "memory": 1, "merkle": 2, "poseidon2": 3, "range_check_8": 4, "range_check_16": 5,
          "range_check_20": 6, "bitwise": 7}
N_REL, MAX_REL = 8, 16
PREPROC_LOG = [18, 18, 18, 18, 8, 16, 20]
# reference column id (PreProcessedColumn::id) -> air::PreprocId; only the four lookup-table components read these columns
PP_INDEX = {"bitwise_stacked_col_0": 0, "bitwise_stacked_col_1": 1, "bitwise_stacked_col_2": 2, "bitwise_stacked_col_3": 3,
            "range_check_8": 4, "range_check_16": 5, "range_check_20": 6}
P = 2**31 - 1
LOG = 4


# ---- M31 / CM31 / QM31 in plain Python (QM31 = (a + b i) + (c + d i) u, i^2 = -1, u^2 = 2 + i) ----
def cmul(x, y):
    return ((x[0] * y[0] - x[1] * y[1]) % P, (x[0] * y[1] + x[1] * y[0]) % P)


def qadd(x, y):
    return tuple((a + b) % P for a, b in zip(x, y))


def qsub(x, y):
    return tuple((a - b) % P for a, b in zip(x, y))


def qmul(x, y):
    a, b, c, d = x[:2], x[2:], y[:2], y[2:]
    ac, bd = cmul(a, c), cmul(b, d)
    r_bd = cmul(bd, (2, 1))
    ad, bc = cmul(a, d), cmul(b, c)
    return ((ac[0] + r_bd[0]) % P, (ac[1] + r_bd[1]) % P, (ad[0] + bc[0]) % P, (ad[1] + bc[1]) % P)


def qscale(x, m):
    return tuple(a * m % P for a in x)


def qfrom(m):
    return (m % P, 0, 0, 0)


# ---- circle group: the two vanishing-polynomial values of CanonicCoset(LOG) on the evaluation domain CanonicCoset(LOG + 1) ----
def cadd(p, q):
    return ((p[0] * q[0] - p[1] * q[1]) % P, (p[0] * q[1] + p[1] * q[0]) % P)


def point_at_index(idx):
    res, cur = (1, 0), (2, 1268011823)
    idx &= 0x7fffffff
    while idx:
        if idx & 1:
            res = cadd(res, cur)
        cur = cadd(cur, cur)
        idx >>= 1
    return res


def denom_inverses(log):
    """1 / coset_vanishing(CanonicCoset(log), p) on the evaluation domain CanonicCoset(log + 1): the vanishing polynomial takes one
    value on the even and another on the odd natural indices of that domain, i.e. (bit-reversed storage) one per HALF of the
    stored rows — the values at natural indices 0 and 1 (domain point i of the first half-coset = G^(init + i * step))"""
    init, step = 1 << (31 - (log + 2)), 1 << (31 - log)
    out = []
    for i in (0, 1):
        x = point_at_index(init + step * i)[0]
        for _ in range(1, log):
            x = (2 * x * x - 1) % P
        out.append(pow(x, P - 2, P))
    return out


def expected_row(g_row, rels_z, rels_apow, coeffs, n_base, dinv, shift=(0, 0, 0, 0)):
    total = (0, 0, 0, 0)
    cons = g_row["constraints"]
    assert len(cons) == n_base
    for k, c in enumerate(cons):
        total = qadd(total, qscale(coeffs[k], c))
    ents = []
    for rel, mult, vals in g_row["relations"]:
        r = REL_ID[rel]
        den = (0, 0, 0, 0)
        for i, v in enumerate(vals):
            den = qadd(den, qscale(rels_apow[r][i], v))
        ents.append((mult % P, qsub(den, rels_z[r])))
    j = 0
    per = 2 if g_row.get("finalize", "pairs") == "pairs" else 1    # finalize_logup_in_pairs / finalize_logup
    n_batches = (len(ents) + per - 1) // per
    for i in range(0, len(ents), per):
        if per == 2 and i + 1 < len(ents):
            (n0, d0), (n1, d1) = ents[i], ents[i + 1]
            num, den = qadd(qscale(d1, n0), qscale(d0, n1)), qmul(d0, d1)
        else:
            num, den = qfrom(ents[i][0]), ents[i][1]
        # zero interaction columns: (c_j - c_{j-1}) * D - N = -N; the LAST batch carries the cumulative-sum shift:
        # (cur - prev_row - prev_col + shift) * D - N = shift * D - N  — which also puts the denominator of a lone last entry
        # (an odd number of relation entries) under test
        term = qsub(qmul(shift, den), num) if j == n_batches - 1 else qsub((0, 0, 0, 0), num)
        total = qadd(total, qmul(coeffs[n_base + j], term))
        j += 1
    return qscale(total, dinv), j


def _names(oracle):
    import ctypes as C
    oracle.L.orc_component_name.restype = C.c_char_p
    return {oracle.L.orc_component_name(C.c_int(c)).decode(): c for c in range(34)}


@pytest.mark.parametrize("name", sorted(GOLD))
def test_hip_domain_eval_matches_reference_derived_vectors(backend, oracle, name):
    cid = _names(oracle)[name]          # (component ids only: the names of the golden file -> air::ComponentId)
    g = GOLD[name]
    n_trace, n_inter, n_cons = backend.component_info(cid)
    assert n_trace == g["n_trace"]
    rows = g["rows"]
    n_eval = 2 << LOG
    assert 0 < len(rows) <= n_eval
    rng = np.random.default_rng(7000 + cid)
    # golden row k sits at evaluation-domain rows k (first half) and n_eval - 1 - k (second half: the other vanishing value)
    place = {}
    for k in range(len(rows)):
        place[k] = k
        if n_eval - 1 - k not in place and n_eval - 1 - k >= len(rows):
            place[n_eval - 1 - k] = k
    tr = np.zeros((n_trace, n_eval), dtype=np.uint32)
    for r, k in place.items():
        tr[:, r] = np.asarray(rows[k]["trace"], dtype=np.uint32)
    rel = rng.integers(1, P, size=(N_REL + N_REL * MAX_REL) * 4, dtype=np.uint32)
    rels_z = [tuple(int(x) for x in rel[4 * r:4 * r + 4]) for r in range(N_REL)]
    ap = rel[4 * N_REL:].reshape(N_REL, MAX_REL, 4)
    rels_apow = [[tuple(int(x) for x in ap[r, i]) for i in range(MAX_REL)] for r in range(N_REL)]
    coeff = rng.integers(0, P, size=4 * n_cons, dtype=np.uint32)
    coeffs = [tuple(int(x) for x in coeff[4 * k:4 * k + 4]) for k in range(n_cons)]
    dinv = denom_inverses(LOG)
    claimed = rng.integers(1, P, size=4, dtype=np.uint32)                     # InteractionClaim: cumsum shift = claimed / 2^LOG
    shift = qscale(tuple(int(x) for x in claimed), pow(1 << LOG, P - 2, P))
    h_tr = [backend.upload(tr[c]) for c in range(n_trace)]
    h_it = [backend.upload(np.zeros(n_eval, dtype=np.uint32)) for _ in range(n_inter)]
    pp = np.zeros((len(PREPROC_LOG), n_eval), dtype=np.uint32)      # read by the four lookup-table components only
    for r, k in place.items():
        for cid_str, v in rows[k].get("preproc", {}).items():
            pp[PP_INDEX[cid_str], r] = v
    h_pp = [backend.upload(pp[i]) for i in range(len(PREPROC_LOG))]
    h_acc = [backend.upload(np.zeros(n_eval, dtype=np.uint32)) for _ in range(4)]
    try:
        backend.constraints_accumulate(cid, h_tr, h_it, h_pp, LOG, rel, coeff, claimed, h_acc)
        acc = np.stack([backend.download(h, n_eval) for h in h_acc])
        for r, k in sorted(place.items()):
            per = 2 if rows[k].get("finalize", "pairs") == "pairs" else 1
            n_batches = (len(rows[k]["relations"]) + per - 1) // per
            want, j = expected_row(rows[k], rels_z, rels_apow, coeffs, n_cons - n_batches, dinv[r >> LOG], shift)
            assert j == n_batches and n_inter == 4 * n_batches
            assert tuple(int(x) for x in acc[:, r]) == want, (name, "evaluation-domain row", r, "golden row", k)
            # the comparison is sensitive: the same row with ONE constraint value or ONE tuple element changed does not match
            bad = json.loads(json.dumps(rows[k]))
            if bad["constraints"]:
                bad["constraints"][-1] = (bad["constraints"][-1] + 1) % P
            else:
                bad["relations"][0][2][0] = (bad["relations"][0][2][0] + 1) % P
            assert expected_row(bad, rels_z, rels_apow, coeffs, n_cons - n_batches, dinv[r >> LOG], shift)[0] != want
            bad = json.loads(json.dumps(rows[k]))
            bad["relations"][-1][2][-1] = (bad["relations"][-1][2][-1] + 1) % P
            assert expected_row(bad, rels_z, rels_apow, coeffs, n_cons - n_batches, dinv[r >> LOG], shift)[0] != want
        assert acc.any()
    finally:
        for h in h_tr + h_it + h_pp + h_acc:
            backend.col_free(h)
