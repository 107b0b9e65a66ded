"""GPU parity (through the C ABI) for the commit path: circle IFFT / LDE, eval_at_point,
Blake2s Merkle layers, grind — bit-exact against the CPU oracle on seeded inputs."""
import numpy as np
import pytest

P = 2**31 - 1
pytestmark = pytest.mark.gpu


def rand_cols(rng, ncols, log_n):
    return [rng.integers(0, P, size=1 << log_n, dtype=np.uint32) for _ in range(ncols)]


@pytest.mark.parametrize("log_n", [1, 2, 4, 5, 10, 11, 12, 13, 16, 18])
def test_interpolate_evaluate_parity(backend, oracle, log_n):
    rng = np.random.default_rng(100 + log_n)
    tw = backend.twiddles(log_n + 1)
    cols = rand_cols(rng, 3, log_n)
    hs = [backend.upload(c) for c in cols]
    backend.interpolate(hs, log_n, tw)
    for h, c in zip(hs, cols):
        got = backend.download(h, 1 << log_n)
        assert np.array_equal(got, oracle.interpolate(c))
    outs = [backend.col_alloc(2 << log_n) for _ in cols]
    backend.evaluate(hs, log_n, log_n + 1, tw, outs)
    for h, o, c in zip(hs, outs, cols):
        coeffs = backend.download(h, 1 << log_n)
        assert np.array_equal(backend.download(o, 2 << log_n), oracle.evaluate(coeffs, log_n + 1))
    # same-size evaluate is the inverse of interpolate (round trip)
    outs2 = [backend.col_alloc(1 << log_n) for _ in cols]
    backend.evaluate(hs, log_n, log_n, tw, outs2)
    for o, c in zip(outs2, cols):
        assert np.array_equal(backend.download(o, 1 << log_n), c)
    for h in hs + outs + outs2:
        backend.col_free(h)
    backend.twiddles_free(tw)


@pytest.mark.parametrize("log_n", [21, 22])
def test_interpolate_evaluate_parity_full_size(backend, oracle, log_n):
    """The sizes the metric config transforms (2^21-row components, 2^22 LDE): one column, IFFT and LDE bit-exact
    against the oracle (covers the 3-pass plans 11-5-5 / 11-6-5 / the fused 2-pass plans of kernels_poly.hip)."""
    rng = np.random.default_rng(900 + log_n)
    tw = backend.twiddles(log_n + 1)
    c = rng.integers(0, P, size=1 << log_n, dtype=np.uint32)
    h = backend.upload(c)
    backend.interpolate([h], log_n, tw)
    coeffs = backend.download(h, 1 << log_n)
    assert np.array_equal(coeffs, oracle.interpolate(c))
    o = backend.col_alloc(2 << log_n)
    backend.evaluate([h], log_n, log_n + 1, tw, [o])
    assert np.array_equal(backend.download(o, 2 << log_n), oracle.evaluate(coeffs, log_n + 1))
    backend.col_free(h)
    backend.col_free(o)
    backend.twiddles_free(tw)


@pytest.mark.parametrize("log_n", [4, 12, 17, 18, 19, 20, 21, 22])
@pytest.mark.parametrize("in_place", [False, True])
def test_interpolate_extend_parity(backend, oracle, log_n, in_place):
    """extend_evals in one call (cm_interpolate_extend): 2^18..2^21 go through the fused sweep (last inverse pass + top layer +
    first forward pass of both halves, k_fft_fused_rb), the other sizes through the two separate transforms; coefficients and
    the extension equal the oracle's interpolate / evaluate word for word, out of place and in place."""
    rng = np.random.default_rng(4000 + log_n)
    ncols = 3 if log_n <= 20 else 1
    tw = backend.twiddles(log_n + 1)
    cols = rand_cols(rng, ncols, log_n)
    ev = [backend.upload(c) for c in cols]
    co = ev if in_place else [backend.col_alloc(1 << log_n) for _ in cols]
    ld = [backend.col_alloc(2 << log_n) for _ in cols]
    backend.interpolate_extend(ev, co, ld, log_n, tw)
    for e, c_h, l_h, c in zip(ev, co, ld, cols):
        want_c = oracle.interpolate(c)
        assert np.array_equal(backend.download(c_h, 1 << log_n), want_c)
        assert np.array_equal(backend.download(l_h, 2 << log_n), oracle.evaluate(want_c, log_n + 1))
        if not in_place:
            assert np.array_equal(backend.download(e, 1 << log_n), c)   # the evaluations are left alone
    for h in set(ev + co + ld):
        backend.col_free(h)
    backend.twiddles_free(tw)


def test_lde_roundtrip_large(backend):
    """Full-size property check (no oracle): 2^22 -> LDE 2^23 -> every even-half restriction
    interpolates back; here: interpolate(evaluate(c, n), n) == c and linearity."""
    log_n = 22
    rng = np.random.default_rng(7)
    tw = backend.twiddles(log_n + 1)
    a = rng.integers(0, P, size=1 << log_n, dtype=np.uint32)
    b = rng.integers(0, P, size=1 << log_n, dtype=np.uint32)
    s = ((a.astype(np.uint64) + b) % P).astype(np.uint32)
    ha, hb, hs_ = backend.upload(a), backend.upload(b), backend.upload(s)
    oa, ob, os_ = (backend.col_alloc(2 << log_n) for _ in range(3))
    backend.evaluate([ha, hb, hs_], log_n, log_n + 1, tw, [oa, ob, os_])
    ea, eb, es = (backend.download(o, 2 << log_n) for o in (oa, ob, os_))
    assert np.array_equal(((ea.astype(np.uint64) + eb) % P).astype(np.uint32), es)
    backend.interpolate([oa], log_n + 1, tw)
    back = backend.download(oa, 2 << log_n)
    assert np.array_equal(back[: 1 << log_n], a) and not back[1 << log_n:].any()
    for h in (ha, hb, hs_, oa, ob, os_):
        backend.col_free(h)
    backend.twiddles_free(tw)


@pytest.mark.parametrize("log_n", [3, 9, 10, 11, 15])
def test_eval_at_point_parity(backend, oracle, log_n):
    rng = np.random.default_rng(5 + log_n)
    cols = rand_cols(rng, 4, log_n)
    hs = [backend.upload(c) for c in cols]
    pt = rng.integers(0, P, size=8, dtype=np.uint32)
    got = backend.eval_at_point(hs, log_n, pt)
    for g, c in zip(got, cols):
        assert np.array_equal(g, oracle.eval_at_point(c, pt))
    for h in hs:
        backend.col_free(h)


@pytest.mark.parametrize("logs", [[6, 6, 6], [8] * 17 + [5] * 3 + [3], [10] * 40 + [9] * 16 + [4] * 5, [1], [12, 3],
                                  # k_merkle_top: multi-block ticket (2^16 = 256 blocks), columns entering at phase-1 and
                                  # phase-2 levels, single-layer kernels above; a wide layer (70 columns) forces the old path
                                  [17] * 3 + [16] * 2 + [13] * 5 + [9] * 2 + [7] * 20 + [5] * 3 + [2], [16] * 4, [9] * 33,
                                  [15] * 2 + [11] * 70 + [6] * 3,
                                  # layers of 2^19 nodes and more go through k_merkle_layer (one node per lane, the dominant
                                  # kernel of the bench): leaf layer without children, 18- and 4-column layers with children
                                  [20] * 3 + [19] * 18 + [12] * 2, [19] * 4, [21] + [20] * 42 + [19]])
def test_merkle_commit_parity(backend, oracle, logs):
    rng = np.random.default_rng(len(logs))
    cols = [rng.integers(0, P, size=1 << l, dtype=np.uint32) for l in logs]
    hs = [backend.upload(c) for c in cols]
    root = backend.merkle_commit(hs, logs)
    want, _ = oracle.merkle_commit(cols)
    assert root == want
    for h in hs:
        backend.col_free(h)


@pytest.mark.parametrize("n_cols", [0, 4, 16, 17, 42])
@pytest.mark.parametrize("with_prev", [False, True])
def test_merkle_commit_layer_parity(backend, oracle, n_cols, with_prev):
    """MerkleOps::commit_on_layer through the C ABI (cm_merkle_commit_layer) at 2^19 nodes = k_merkle_layer: every hash of
    the layer equals the oracle's, with and without a previous layer, for column counts below / at / above one 16-word
    Blake2s chunk and for the 42-column shape of the metric config's largest component."""
    if n_cols == 0 and not with_prev:
        pytest.skip("a layer needs children or columns")
    L = 19
    rng = np.random.default_rng(1000 + n_cols + (100 if with_prev else 0))
    upper = [rng.integers(0, P, size=2 << L, dtype=np.uint32) for _ in range(3)] if with_prev else []
    here = [rng.integers(0, P, size=1 << L, dtype=np.uint32) for _ in range(n_cols)]
    _, layers = oracle.merkle_commit(upper + here)
    layers = layers.reshape(-1, 8)
    prev = 0
    hs = []
    if with_prev:
        hu = [backend.upload(c) for c in upper]
        prev = backend.col_alloc(8 * (2 << L))
        backend.merkle_commit_layer(L + 1, 0, hu, prev)
        assert np.array_equal(backend.download(prev, 8 * (2 << L)).reshape(-1, 8), layers[: 2 << L])
        hs += hu + [prev]
        want = layers[2 << L: (2 << L) + (1 << L)]
    else:
        want = layers[: 1 << L]
    hc = [backend.upload(c) for c in here]
    out = backend.col_alloc(8 << L)
    backend.merkle_commit_layer(L, prev, hc, out)
    got = backend.download(out, 8 << L).reshape(-1, 8)
    assert np.array_equal(got, want)
    for h in hs + hc + [out]:
        backend.col_free(h)


def test_grind_parity(backend, oracle):
    rng = np.random.default_rng(3)
    for bits in (2, 8, 16):
        digest = bytes(rng.integers(0, 256, size=32, dtype=np.uint8))
        assert backend.grind(digest, bits) == oracle.grind(digest, bits)


def test_column_element_access_and_copy(backend):
    """Column::at / Column::set / Column::clone of a Rust-side backend (integration/prover-hip/src/backend.rs): cm_col_read,
    cm_col_write, cm_col_copy."""
    rng = np.random.default_rng(5)
    a = rng.integers(0, 2**31 - 1, size=1000, dtype=np.uint32)
    h = backend.upload(a)
    assert backend.col_read(h, 17, 5).tolist() == a[17:22].tolist()
    backend.col_write(h, 998, np.array([7, 9], dtype=np.uint32))
    a[998:] = [7, 9]
    g = backend.col_alloc(1000)
    backend.col_copy(g, h, 1000)
    assert np.array_equal(backend.download(g, 1000), a)
    backend.col_free(h)
    backend.col_free(g)
