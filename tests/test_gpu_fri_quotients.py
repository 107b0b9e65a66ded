"""GPU parity (through the C ABI) for the FieldOps / FriOps / QuotientOps entry points:
cm_batch_inverse_*, cm_fri_fold_circle_into_line, cm_fri_fold_line, cm_accumulate_quotients — bit-exact
against the CPU oracle (oracle/oprover.hpp: accumulate_quotients, fold_circle_into_line, fold_line)."""
import numpy as np
import pytest

P = 2**31 - 1
pytestmark = pytest.mark.gpu


def rand_secure(rng, n):
    return [rng.integers(0, P, size=n, dtype=np.uint32) for _ in range(4)]


def test_batch_inverse_m31(backend, oracle):
    rng = np.random.default_rng(7)
    a = rng.integers(1, P, size=5000, dtype=np.uint32)
    a[:3] = [1, P - 1, 2]
    h, o = backend.upload(a), backend.col_alloc(a.size)
    backend.batch_inverse_m31(h, o, a.size)
    assert np.array_equal(backend.download(o, a.size), oracle.m31_inv(a))
    backend.col_free(h); backend.col_free(o)


def test_batch_inverse_qm31(backend, oracle):
    rng = np.random.default_rng(8)
    n = 3001
    cols = rand_secure(rng, n)
    cols[1][0] = cols[2][0] = cols[3][0] = 0          # base-field element embedded in QM31
    hs = [backend.upload(c) for c in cols]
    os_ = [backend.col_alloc(n) for _ in range(4)]
    backend.batch_inverse_qm31(hs, os_, n)
    got = np.stack([backend.download(o, n) for o in os_], axis=1)
    exp = oracle.qm31_inv(np.stack(cols, axis=1).reshape(-1)).reshape(n, 4)
    assert np.array_equal(got, exp)
    for h in hs + os_:
        backend.col_free(h)


@pytest.mark.parametrize("log_n", [2, 3, 8, 12, 15])
def test_fold_circle_into_line(backend, oracle, log_n):
    rng = np.random.default_rng(20 + log_n)
    tw = backend.twiddles(log_n)
    src = rand_secure(rng, 1 << log_n)
    dst = rand_secure(rng, 1 << (log_n - 1))
    alpha = rng.integers(0, P, size=4, dtype=np.uint32)
    hs, hd = [backend.upload(c) for c in src], [backend.upload(c) for c in dst]
    backend.fri_fold_circle_into_line(hd, hs, alpha, log_n, tw)
    got = np.stack([backend.download(h, 1 << (log_n - 1)) for h in hd])
    assert np.array_equal(got, oracle.fold_circle_into_line(dst, src, log_n, alpha))
    for h in hs + hd:
        backend.col_free(h)
    backend.twiddles_free(tw)


@pytest.mark.parametrize("log_n", [1, 2, 7, 12, 15])
def test_fold_line(backend, oracle, log_n):
    rng = np.random.default_rng(40 + log_n)
    tw = backend.twiddles(log_n + 1)
    src = rand_secure(rng, 1 << log_n)
    alpha = rng.integers(0, P, size=4, dtype=np.uint32)
    hs = [backend.upload(c) for c in src]
    ho = [backend.col_alloc(1 << (log_n - 1)) for _ in range(4)]
    backend.fri_fold_line(hs, alpha, log_n, tw, ho)
    got = np.stack([backend.download(h, 1 << (log_n - 1)) for h in ho])
    assert np.array_equal(got, oracle.fold_line(src, log_n, alpha))
    for h in hs + ho:
        backend.col_free(h)
    backend.twiddles_free(tw)


@pytest.mark.parametrize("mode", ["line", "line+circle", "circle"])
@pytest.mark.parametrize("log_n", [5, 15, 20, 21])
def test_fold_line_leaves(backend, oracle, log_n, mode):
    """cm_fri_fold_line_leaves (k_fold_leaf: the fold of a FRI layer and the leaf layer of its Merkle tree in one pass; log_n < 15
    takes the two separate launches): the folded layer equals the oracle's fold_line (with the quotient columns of that size
    folded in: fold_circle_into_line accumulating with the circle challenge; or the circle fold alone into a blank layer, the
    first inner layer), and every leaf hash equals the oracle's commitment layer over the four coordinate columns.
    2^20 / 2^21 values: chunk counts 1 and 2 per wave."""
    rng = np.random.default_rng(700 + log_n + 50 * ["line", "line+circle", "circle"].index(mode))
    n_out = 1 << (log_n - 1)
    tw = backend.twiddles(log_n + 1)
    hs, hq, alpha, ac = None, None, None, None
    want = np.zeros((4, n_out), dtype=np.uint32)
    if mode != "circle":
        src = rand_secure(rng, 1 << log_n)
        alpha = rng.integers(0, P, size=4, dtype=np.uint32)
        want = oracle.fold_line(src, log_n, alpha)
        hs = [backend.upload(c) for c in src]
    if mode != "line":
        circ = rand_secure(rng, 1 << log_n)
        ac = rng.integers(0, P, size=4, dtype=np.uint32)
        want = oracle.fold_circle_into_line(want, circ, log_n, ac)   # dst * ac^2 + fold_circle(circ, ac); dst = 0 when alone
        hq = [backend.upload(c) for c in circ]
    ho = [backend.col_alloc(n_out) for _ in range(4)]
    hh = backend.col_alloc(8 * n_out)
    backend.fri_fold_line_leaves(hs, alpha, log_n, tw, ho, hh, circle4=hq, alpha_circle=ac)
    got = np.stack([backend.download(h, n_out) for h in ho])
    assert np.array_equal(got, want)
    _, layers = oracle.merkle_commit([np.ascontiguousarray(want[k]) for k in range(4)])
    assert np.array_equal(backend.download(hh, 8 * n_out).reshape(-1, 8), layers.reshape(-1, 8)[:n_out])
    for h in (hs or []) + (hq or []) + ho + [hh]:
        backend.col_free(h)
    backend.twiddles_free(tw)


@pytest.mark.parametrize("log_n,n_cols", [(3, 1), (6, 5), (11, 40), (14, 7)])
def test_accumulate_quotients(backend, oracle, log_n, n_cols):
    """Two sample points (the OODS point and its mask-shifted neighbour): every column is sampled at point 0,
    every third column also at point 1 — the shape Cairo-M's interaction columns produce."""
    rng = np.random.default_rng(60 + log_n)
    tw = backend.twiddles(log_n)
    cols = [rng.integers(0, P, size=1 << log_n, dtype=np.uint32) for _ in range(n_cols)]
    points = rng.integers(0, P, size=(2, 8), dtype=np.uint32)
    b0 = list(range(n_cols))
    b1 = list(range(0, n_cols, 3))
    col_index = np.array(b0 + b1, dtype=np.uint32)
    batch_off = np.array([0, len(b0), len(b0) + len(b1)], dtype=np.uint32)
    values = rng.integers(0, P, size=(col_index.size, 4), dtype=np.uint32)
    coeff = rng.integers(0, P, size=4, dtype=np.uint32)
    hs = [backend.upload(c) for c in cols]
    ho = [backend.col_alloc(1 << log_n) for _ in range(4)]
    backend.accumulate_quotients(log_n, hs, points, batch_off, col_index, values, coeff, ho, tw)
    got = np.stack([backend.download(h, 1 << log_n) for h in ho])
    exp = oracle.accumulate_quotients(log_n, cols, points, batch_off, col_index, values, coeff)
    assert np.array_equal(got, exp)
    for h in hs + ho:
        backend.col_free(h)
    backend.twiddles_free(tw)
