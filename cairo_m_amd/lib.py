"""ctypes binding of libcairom_hip.so (C ABI in include/cairom_hip.h).

Mirrors, op by op, the Stwo backend-trait calls the reference prover makes
(/root/reference/crates/prover/src/prover.rs:56-131): twiddles, interpolate, evaluate,
eval_at_point, Merkle commit, grind, and the whole-segment ``prove``.
"""
import ctypes as C
import os
import numpy as np

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcairom_hip.so")
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)


class CmError(RuntimeError):
    pass


def load_library(path=LIB_PATH):
    if not os.path.exists(path):
        raise CmError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(no CPU fallback exists)")
    return C.CDLL(path)


def _p(a):
    return a.ctypes.data_as(_u32p)


class Backend:
    """Host-side handle on the HIP backend.  One instance per process / GPU."""

    def __init__(self, device=0, path=LIB_PATH):
        self.L = load_library(path)
        self.L.cm_last_error.restype = C.c_int32
        self._ck(self.L.cm_init(C.c_int32(device)))

    # -- plumbing ------------------------------------------------------------------------
    def _ck(self, rc):
        if rc != 0:
            buf = C.create_string_buffer(1024)
            self.L.cm_last_error(buf, C.c_size_t(1024))
            raise CmError(f"libcairom_hip status {rc}: {buf.value.decode(errors='replace')}")

    def col_alloc(self, n):
        h = C.c_uint64(0)
        self._ck(self.L.cm_col_alloc(C.c_uint64(n), C.byref(h)))
        return h.value

    def col_free(self, h):
        self._ck(self.L.cm_col_free(C.c_uint64(h)))

    def upload(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.uint32)
        h = self.col_alloc(arr.size)
        self._ck(self.L.cm_col_h2d(C.c_uint64(h), _p(arr), C.c_uint64(arr.size), C.c_uint64(0)))
        return h

    def download(self, h, n):
        out = np.empty(n, dtype=np.uint32)
        self._ck(self.L.cm_col_d2h(C.c_uint64(h), _p(out), C.c_uint64(n), C.c_uint64(0)))
        return out

    @staticmethod
    def _harr(hs):
        return (C.c_uint64 * len(hs))(*hs)

    # -- PolyOps ----------------------------------------------------------------------------
    def twiddles(self, log_size):
        h = C.c_uint64(0)
        self._ck(self.L.cm_twiddles_precompute(C.c_uint32(log_size), C.byref(h)))
        return h.value

    def twiddles_free(self, tw):
        self._ck(self.L.cm_twiddles_free(C.c_uint64(tw)))

    def interpolate(self, cols, log_n, tw):
        self._ck(self.L.cm_interpolate(self._harr(cols), C.c_uint32(len(cols)), C.c_uint32(log_n), C.c_uint64(tw),
                                       C.c_uint64(0)))

    def evaluate(self, coeffs, log_n, log_out, tw, out):
        self._ck(self.L.cm_evaluate(self._harr(coeffs), C.c_uint32(len(coeffs)), C.c_uint32(log_n),
                                    C.c_uint32(log_out), C.c_uint64(tw), self._harr(out), C.c_uint64(0)))

    def bit_reverse(self, cols, log_n):
        self._ck(self.L.cm_bit_reverse(self._harr(cols), C.c_uint32(len(cols)), C.c_uint32(log_n), C.c_uint64(0)))

    def eval_at_point(self, coeffs, log_n, pt_xy):
        pt = np.ascontiguousarray(pt_xy, dtype=np.uint32)
        out = np.empty(4 * len(coeffs), dtype=np.uint32)
        self._ck(self.L.cm_eval_at_point(self._harr(coeffs), C.c_uint32(len(coeffs)), C.c_uint32(log_n), _p(pt),
                                         _p(out), C.c_uint64(0)))
        return out.reshape(-1, 4)

    # -- MerkleOps / GrindOps ---------------------------------------------------------------
    def merkle_commit(self, cols, col_logs):
        logs = np.ascontiguousarray(col_logs, dtype=np.uint32)
        root = (C.c_uint8 * 32)()
        self._ck(self.L.cm_merkle_commit(self._harr(cols), _p(logs), C.c_uint32(len(cols)), root, C.c_uint64(0)))
        return bytes(root)

    def merkle_commit_layer(self, log_size, prev, cols, out):
        self._ck(self.L.cm_merkle_commit_layer(C.c_uint32(log_size), C.c_uint64(prev), self._harr(cols),
                                               C.c_uint32(len(cols)), C.c_uint64(out), C.c_uint64(0)))

    def grind(self, digest, bits):
        d = (C.c_uint8 * 32)(*digest)
        nonce = C.c_uint64(0)
        self._ck(self.L.cm_grind(d, C.c_uint32(bits), C.byref(nonce)))
        return nonce.value
