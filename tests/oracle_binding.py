"""ctypes binding of oracle/liboracle.so — the CPU restatement used as the parity checker."""
import ctypes as C
import os
import numpy as np

_u32p = C.POINTER(C.c_uint32)


def _p(a):
    return a.ctypes.data_as(_u32p)


def oracle_threads():
    """OpenMP threads the oracle runs with: its loops are short, so beyond a few tens of threads the
    fork/join cost dominates (a 256-core host is slower than an 8-core one)."""
    return int(os.environ.get("ORACLE_THREADS", min(16, os.cpu_count() or 1)))


class Oracle:
    def __init__(self, so):
        os.environ.setdefault("OMP_NUM_THREADS", str(oracle_threads()))  # read by libgomp when it loads
        self.L = C.CDLL(so)
        self.L.orc_grind.restype = C.c_uint64

    def interpolate(self, vals):
        v = np.ascontiguousarray(vals, dtype=np.uint32).copy()
        self.L.orc_interpolate(_p(v), C.c_uint32(int(np.log2(v.size))))
        return v

    def evaluate(self, coeffs, log_out):
        c = np.ascontiguousarray(coeffs, dtype=np.uint32)
        out = np.empty(1 << log_out, dtype=np.uint32)
        self.L.orc_evaluate(_p(c), C.c_uint32(int(np.log2(c.size))), _p(out), C.c_uint32(log_out))
        return out

    def eval_at_point(self, coeffs, pt_xy):
        c = np.ascontiguousarray(coeffs, dtype=np.uint32)
        pt = np.ascontiguousarray(pt_xy, dtype=np.uint32)
        out = np.empty(4, dtype=np.uint32)
        self.L.orc_eval_at_point(_p(c), C.c_uint32(int(np.log2(c.size))), _p(pt), _p(out))
        return out

    def domain_point(self, log, i):
        xy = np.zeros(2, dtype=np.uint32)
        self.L.orc_domain_point(C.c_uint32(log), C.c_uint64(i), _p(xy))
        return int(xy[0]), int(xy[1])

    def blake2s(self, data):
        out = (C.c_uint8 * 32)()
        self.L.orc_blake2s256(data, C.c_size_t(len(data)), out)
        return bytes(out)

    def merkle_commit(self, cols):
        """cols: list of 1-D uint32 arrays (power-of-two lengths). Returns (root, layers bytes)."""
        data = np.concatenate([np.ascontiguousarray(c, dtype=np.uint32) for c in cols])
        logs = np.array([int(np.log2(c.size)) for c in cols], dtype=np.uint32)
        mx = int(logs.max())
        root = (C.c_uint8 * 32)()
        layers = np.zeros(((2 << mx) - 1) * 8, dtype=np.uint32)
        self.L.orc_merkle_commit(_p(data), _p(logs), C.c_size_t(len(cols)), root,
                                 layers.ctypes.data_as(C.POINTER(C.c_uint8)))
        return bytes(root), layers

    def grind(self, digest, bits):
        d = (C.c_uint8 * 32)(*digest)
        return int(self.L.orc_grind(d, C.c_uint32(bits)))

    def m31_mul(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.uint32); b = np.ascontiguousarray(b, dtype=np.uint32)
        o = np.empty_like(a)
        self.L.orc_m31_mul(_p(a), _p(b), _p(o), C.c_size_t(a.size))
        return o

    def m31_inv(self, a):
        a = np.ascontiguousarray(a, dtype=np.uint32)
        o = np.empty_like(a)
        self.L.orc_m31_inv(_p(a), _p(o), C.c_size_t(a.size))
        return o

    def qm31_mul(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.uint32); b = np.ascontiguousarray(b, dtype=np.uint32)
        o = np.empty_like(a)
        self.L.orc_qm31_mul(_p(a), _p(b), _p(o), C.c_size_t(a.size // 4))
        return o

    def qm31_inv(self, a):
        a = np.ascontiguousarray(a, dtype=np.uint32)
        o = np.empty_like(a)
        self.L.orc_qm31_inv(_p(a), _p(o), C.c_size_t(a.size // 4))
        return o


def _attach_prover(cls):
    def setup(self):
        self.L.orc_last_error.restype = C.c_char_p
        self.L.orc_proof_n_words.restype = C.c_uint64
        self.L.orc_proof_cells.restype = C.c_uint64

    def err(self):
        return self.L.orc_last_error().decode(errors="replace")

    def prove(self, view, cfg=(16, 1, 0, 80), transcript=False):
        """CPU restatement of prove_cairo_m; returns (words, cells) — or (words, cells, transcript entries)."""
        setup(self)
        h = C.c_void_p()
        rc = self.L.orc_prove(view, (C.c_uint32 * 4)(*cfg), C.byref(h))
        if rc:
            raise RuntimeError("oracle prove failed: " + err(self))
        n = self.L.orc_proof_n_words(h)
        w = np.zeros(n, dtype=np.uint32)
        self.L.orc_proof_words(h, _p(w))
        cells = self.L.orc_proof_cells(h)
        tr = None
        if transcript:
            import json
            self.L.orc_proof_transcript.restype = C.c_char_p
            tr = json.loads(self.L.orc_proof_transcript(h).decode())
        self.L.orc_proof_free(h)
        return (w, cells, tr) if transcript else (w, cells)

    def set_framing(self, spec):
        """orc_set_framing: the oracle's copy of the named framing switches (oracle/oframing.hpp)."""
        setup(self)
        if self.L.orc_set_framing((spec or "").encode()):
            raise RuntimeError(err(self))

    def verify(self, words, cfg=None):
        """cfg = the PcsConfig the VERIFIER expects (None = REGULAR_96_BITS), like verify_cairo_m's second argument."""
        setup(self)
        w = np.ascontiguousarray(words, dtype=np.uint32)
        rc = self.L.orc_verify(_p(w), C.c_uint64(w.size), (C.c_uint32 * 4)(*cfg) if cfg else None)
        return rc, (err(self) if rc else "")

    def assert_constraints(self, view):
        setup(self)
        rc = self.L.orc_assert_constraints(view)
        return rc, (err(self) if rc else "")

    def component_trace(self, view, cid):
        setup(self)
        log, ncols = C.c_uint32(0), C.c_uint32(0)
        self.L.orc_component_trace(view, C.c_int(cid), C.byref(log), C.byref(ncols), None, C.c_uint64(0))
        out = np.zeros(ncols.value << log.value, dtype=np.uint32)
        rc = self.L.orc_component_trace(view, C.c_int(cid), C.byref(log), C.byref(ncols), _p(out), C.c_uint64(out.size))
        if rc:
            raise RuntimeError(err(self))
        return out.reshape(ncols.value, 1 << log.value)

    def component_interaction(self, view, cid, rel_words, n_cols, log):
        setup(self)
        out = np.zeros(n_cols << log, dtype=np.uint32)
        cs = np.zeros(4, dtype=np.uint32)
        r = np.ascontiguousarray(rel_words, dtype=np.uint32)
        rc = self.L.orc_component_interaction(view, C.c_int(cid), _p(r), _p(out), C.c_uint64(out.size), _p(cs))
        if rc:
            raise RuntimeError(err(self))
        return out.reshape(n_cols, 1 << log), cs

    def component_constraints(self, view, cid, rel_words, coeff_words, log):
        setup(self)
        out = np.zeros(4 << (log + 1), dtype=np.uint32)
        r = np.ascontiguousarray(rel_words, dtype=np.uint32)
        c = np.ascontiguousarray(coeff_words, dtype=np.uint32)
        rc = self.L.orc_component_constraints(view, C.c_int(cid), _p(r), _p(c), _p(out))
        if rc:
            raise RuntimeError(err(self))
        return out.reshape(4, -1)

    def fri_decompose(self, f4, log):
        f = np.ascontiguousarray(np.concatenate(f4), dtype=np.uint32).copy()
        lam = np.zeros(4, dtype=np.uint32)
        self.L.orc_fri_decompose(_p(f), C.c_uint32(log), _p(lam))
        return f.reshape(4, -1), lam

    def fold_circle_into_line(self, dst4, src4, log, alpha):
        d = np.ascontiguousarray(np.concatenate(dst4), dtype=np.uint32).copy()
        s = np.ascontiguousarray(np.concatenate(src4), dtype=np.uint32)
        a = np.ascontiguousarray(alpha, dtype=np.uint32)
        self.L.orc_fold_circle_into_line(_p(d), _p(s), C.c_uint32(log), _p(a))
        return d.reshape(4, -1)

    def fold_line(self, src4, log, alpha):
        s = np.ascontiguousarray(np.concatenate(src4), dtype=np.uint32)
        a = np.ascontiguousarray(alpha, dtype=np.uint32)
        out = np.empty(4 << (log - 1), dtype=np.uint32)
        self.L.orc_fold_line(_p(s), C.c_uint32(log), _p(a), _p(out))
        return out.reshape(4, -1)

    def accumulate_quotients(self, log, cols, points, batch_off, col_index, values, coeff):
        c = np.ascontiguousarray(np.concatenate(cols), dtype=np.uint32)
        pts = np.ascontiguousarray(points, dtype=np.uint32)
        off = np.ascontiguousarray(batch_off, dtype=np.uint32)
        ci = np.ascontiguousarray(col_index, dtype=np.uint32)
        vals = np.ascontiguousarray(values, dtype=np.uint32)
        co = np.ascontiguousarray(coeff, dtype=np.uint32)
        out = np.empty(4 << log, dtype=np.uint32)
        self.L.orc_accumulate_quotients(C.c_uint32(log), _p(c), C.c_uint32(len(cols)), C.c_uint32(len(off) - 1), _p(pts),
                                        _p(off), _p(ci), _p(vals), _p(co), _p(out))
        return out.reshape(4, -1)

    def poseidon2_permute(self, state):
        s = np.ascontiguousarray(state, dtype=np.uint32).copy()
        self.L.orc_poseidon2_permute(_p(s))
        return s

    cls.prove, cls.verify, cls.assert_constraints, cls.set_framing = prove, verify, assert_constraints, set_framing
    cls.component_trace, cls.poseidon2_permute = component_trace, poseidon2_permute
    cls.component_interaction, cls.component_constraints, cls.fri_decompose = component_interaction, component_constraints, fri_decompose
    cls.fold_circle_into_line, cls.fold_line, cls.accumulate_quotients = fold_circle_into_line, fold_line, accumulate_quotients


_attach_prover(Oracle)
