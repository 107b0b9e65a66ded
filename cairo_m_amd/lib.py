"""ctypes binding of libcairom_hip.so (C ABI in include/cairom_hip.h).

Mirrors, op by op, the Stwo backend-trait calls the reference prover makes
(/root/reference/crates/prover/src/prover.rs:56-131): twiddles, interpolate, evaluate,
eval_at_point, Merkle commit, grind, and the whole-segment ``prove``.
"""
import ctypes as C
import os
import numpy as np

# CAIROM_HIP_LIB: development override (A/B timing of two builds inside one GPU session); the shipped path is in-tree
LIB_PATH = os.environ.get("CAIROM_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcairom_hip.so")
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

    def col_read(self, h, offset, n):
        out = np.empty(n, dtype=np.uint32)
        self._ck(self.L.cm_col_read(C.c_uint64(h), C.c_uint64(offset), _p(out), C.c_uint64(n), C.c_uint64(0)))
        return out

    def col_write(self, h, offset, arr):
        arr = np.ascontiguousarray(arr, dtype=np.uint32)
        self._ck(self.L.cm_col_write(C.c_uint64(h), C.c_uint64(offset), _p(arr), C.c_uint64(arr.size), C.c_uint64(0)))

    def col_copy(self, dst, src, n):
        self._ck(self.L.cm_col_copy(C.c_uint64(dst), C.c_uint64(src), C.c_uint64(n), C.c_uint64(0)))

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

    def interpolate_extend(self, evals, coeffs, lde, log_n, tw):
        """extend_evals at blowup 1: evals -> coeffs (2^log_n) and lde (2^(log_n + 1)); evals may be the coeffs handles"""
        self._ck(self.L.cm_interpolate_extend(self._harr(evals), self._harr(coeffs), self._harr(lde), C.c_uint32(len(evals)),
                                              C.c_uint32(log_n), C.c_uint64(tw), C.c_uint64(0)))

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
        self._ck(self.L.cm_grind(d, C.c_uint32(bits), C.byref(nonce), C.c_uint64(0)))
        return nonce.value

    # -- FieldOps / FriOps / QuotientOps ----------------------------------------------------
    def batch_inverse_m31(self, src, dst, n):
        self._ck(self.L.cm_batch_inverse_m31(C.c_uint64(src), C.c_uint64(dst), C.c_uint64(n), C.c_uint64(0)))

    def batch_inverse_qm31(self, src4, dst4, n):
        self._ck(self.L.cm_batch_inverse_qm31(self._harr(src4), self._harr(dst4), C.c_uint64(n), C.c_uint64(0)))

    def fri_fold_circle_into_line(self, dst4, src4, alpha, log_n, tw):
        a = np.ascontiguousarray(alpha, dtype=np.uint32)
        self._ck(self.L.cm_fri_fold_circle_into_line(self._harr(dst4), self._harr(src4), _p(a), C.c_uint32(log_n),
                                                     C.c_uint64(tw), C.c_uint64(0)))

    def fri_fold_line(self, src4, alpha, log_n, tw, out4):
        a = np.ascontiguousarray(alpha, dtype=np.uint32)
        self._ck(self.L.cm_fri_fold_line(self._harr(src4), _p(a), C.c_uint32(log_n), C.c_uint64(tw),
                                         self._harr(out4), C.c_uint64(0)))

    def fri_fold_line_leaves(self, src4, alpha, log_n, tw, out4, leaf_hashes, circle4=None, alpha_circle=None):
        """A FRI layer (fold_line of src4, fold_circle of circle4 accumulated in; either may be None) and the leaf hashes of the
        folded layer in one pass."""
        a = None if alpha is None else np.ascontiguousarray(alpha, dtype=np.uint32)
        ac = None if alpha_circle is None else np.ascontiguousarray(alpha_circle, dtype=np.uint32)
        self._ck(self.L.cm_fri_fold_line_leaves(None if src4 is None else self._harr(src4), None if circle4 is None else self._harr(circle4),
                                                None if a is None else _p(a), None if ac is None else _p(ac), C.c_uint32(log_n),
                                                C.c_uint64(tw), self._harr(out4), C.c_uint64(leaf_hashes), C.c_uint64(0)))

    def accumulate_quotients(self, log_size, cols, points, batch_off, col_index, values, coeff, out4, tw):
        """points: (n_batches, 8) u32; batch_off: n_batches+1; col_index: entries; values: (entries, 4)."""
        class Batches(C.Structure):
            _fields_ = [("n_batches", C.c_uint32), ("points", C.c_void_p), ("batch_off", C.c_void_p),
                        ("col_index", C.c_void_p), ("values", C.c_void_p)]
        pts = np.ascontiguousarray(points, dtype=np.uint32)
        off = np.ascontiguousarray(batch_off, dtype=np.uint32)
        ci = np.ascontiguousarray(col_index, dtype=np.uint32)
        vals = np.ascontiguousarray(values, dtype=np.uint32)
        co = np.ascontiguousarray(coeff, dtype=np.uint32)
        b = Batches(len(off) - 1, pts.ctypes.data, off.ctypes.data, ci.ctypes.data, vals.ctypes.data)
        self._ck(self.L.cm_accumulate_quotients(C.c_uint32(log_size), self._harr(cols), C.c_uint32(len(cols)), C.byref(b),
                                                _p(co), self._harr(out4), C.c_uint64(tw), C.c_uint64(0)))


class HostInput:
    """ProverInput built on the host by the synthetic VM + adapter (no GPU needed)."""

    def __init__(self, lib, handle):
        self.L = lib
        self.h = handle
        self.L.cm_host_input_view.restype = C.c_void_p
        self.L.cm_host_input_steps.restype = C.c_uint64

    @property
    def view(self):
        return C.c_void_p(self.L.cm_host_input_view(self.h))

    @property
    def steps(self):
        return int(self.L.cm_host_input_steps(self.h))

    def free(self):
        if self.h:
            self.L.cm_host_input_free(self.h)
            self.h = None


def _lib_error(L, rc):
    buf = C.create_string_buffer(2048)
    L.cm_last_error(buf, C.c_size_t(2048))
    return CmError(f"libcairom_hip status {rc}: {buf.value.decode(errors='replace')}")


def synth_fibonacci(n, max_steps=1 << 30, segment=0, lib=None):
    """fibonacci_loop(n) through the synthetic VM + adapter: 10*n + 12 steps."""
    L = lib or load_library()
    h = C.c_void_p()
    rc = L.cm_synth_fibonacci(C.c_uint32(n), C.c_uint64(max_steps), C.c_uint32(segment), C.byref(h))
    if rc:
        raise _lib_error(L, rc)
    return HostInput(L, h)


def vm_run(program, entry_pc=0, args=(), n_returns=0, max_steps=1 << 30, segment=0, lib=None):
    """Run a CASM program (list of instruction word lists) and adapt one segment."""
    L = lib or load_library()
    words = np.array([w for ins in program for w in ins], dtype=np.uint32)
    lens = np.array([len(ins) for ins in program], dtype=np.uint32)
    a = np.array(list(args), dtype=np.uint32)
    h = C.c_void_p()
    nseg = C.c_uint32(0)
    rc = L.cm_vm_run(_p(words), _p(lens), C.c_uint32(len(program)), C.c_uint32(entry_pc), _p(a), C.c_uint32(len(a)),
                     C.c_uint32(n_returns), C.c_uint64(max_steps), C.c_uint32(segment), C.byref(h), C.byref(nseg))
    if rc:
        raise _lib_error(L, rc)
    hi = HostInput(L, h)
    hi.n_segments = nseg.value
    return hi


class Proof:
    PHASES = ["setup", "preprocessed", "trace_gen", "trace_commit", "interaction_gen", "interaction_commit",
              "constraints", "composition_commit", "oods_sampling", "quotients", "fri_commit", "pow", "decommit"]

    def __init__(self, lib, handle):
        self.L = lib
        self.h = handle

    def words(self):
        p = C.POINTER(C.c_uint32)()
        n = C.c_uint64(0)
        rc = self.L.cm_proof_words(self.h, C.byref(p), C.byref(n))
        if rc:
            raise _lib_error(self.L, rc)
        return np.ctypeslib.as_array(p, shape=(n.value,)).copy()

    def json(self):
        p = C.c_char_p()
        n = C.c_size_t(0)
        rc = self.L.cm_proof_json(self.h, C.byref(p), C.byref(n))
        if rc:
            raise _lib_error(self.L, rc)
        return C.string_at(p, n.value).decode()

    def commitments(self):
        roots = ((C.c_uint8 * 32) * 4)()
        self.L.cm_proof_commitments(self.h, roots)
        return [bytes(r) for r in roots]

    def stats(self):
        cells, steps = C.c_uint64(0), C.c_uint64(0)
        ph = (C.c_double * 32)()
        n = self.L.cm_proof_stats(self.h, C.byref(cells), C.byref(steps), ph, C.c_uint32(32))
        return {"cells": cells.value, "steps": steps.value,
                "phase_ms": dict(zip(self.PHASES, [ph[i] for i in range(min(n, 32))]))}

    def verify(self, cfg=None):
        """verify_cairo_m(proof, pcs_config) on this proof (product-side verifier, host code): (status, message).
        cfg = the PcsConfig the verifier expects, None = REGULAR_96_BITS (never taken from the proof)."""
        rc = self.L.cm_verify_proof(self.h, _cfg(cfg))
        buf = C.create_string_buffer(512)
        self.L.cm_last_error(buf, C.c_size_t(512))
        return rc, buf.value.decode(errors="replace") if rc else ""

    def transcript(self):
        """cm_proof_transcript: the Fiat-Shamir steps of this proof (list of {"op", "digest", "n_words", "words"}); empty
        unless set_transcript_log(True) was in force when it was made."""
        import json
        p = C.c_char_p()
        n = C.c_size_t(0)
        rc = self.L.cm_proof_transcript(self.h, C.byref(p), C.byref(n))
        if rc:
            raise _lib_error(self.L, rc)
        return json.loads(C.string_at(p, n.value).decode())

    def free(self):
        if self.h:
            self.L.cm_proof_free(self.h)
            self.h = None


def set_framing(spec, lib=None):
    """cm_set_framing (process-wide; host code, works without a GPU): named switches for the Stwo-side conventions no
    reference vector settles — include/cairom_hip.h.  "" restores the defaults."""
    L = lib or load_library()
    rc = L.cm_set_framing((spec or "").encode())
    if rc:
        raise _lib_error(L, rc)


def get_framing(lib=None):
    L = lib or load_library()
    buf = C.create_string_buffer(256)
    L.cm_get_framing(buf, C.c_size_t(256))
    return buf.value.decode()


def set_transcript_log(on, lib=None):
    (lib or load_library()).cm_set_transcript_log(C.c_int32(1 if on else 0))


class ArrayInput:
    """A ProverInput given as explicit arrays (hand-built inputs such as crates/prover/tests/prover.rs:33-112, or the `input`
    object of a tests/golden/ref_*.json file): the same `.view` / `.steps` / `.free()` surface as HostInput.
    arrays: dict with regs[4], roots[2], ranges[6], bundles<k> (n x 12), data_accesses (n x 4), initial_memory / final_memory
    (n x 7: address, v0..v3, clock, multiplicity — IN THE ROW ORDER the memory component must use), clock_updates (n x 6),
    initial_tree / final_tree (n x 8) — the layout prover_input_arrays() returns."""

    def __init__(self, arrays):
        self._keep = []
        v = ProverInputView()

        def arr(name, words):
            a = np.ascontiguousarray(np.array(arrays.get(name, []), dtype=np.uint32).reshape(-1, words))
            self._keep.append(a)
            return a

        for i, x in enumerate(arrays.get("regs", [0, 0, 0, 0])):
            v.regs[i] = int(x)
        total = 0
        for i in range(N_OPCODE_COMPONENTS):
            a = arr(f"bundles{i}", 12)
            v.bundles[i] = a.ctypes.data if a.shape[0] else None
            v.n_bundles[i] = a.shape[0]
            total += a.shape[0]
        for name, words, fld in (("data_accesses", 4, "data_accesses"), ("initial_memory", 7, "initial_memory"),
                                 ("final_memory", 7, "final_memory"), ("clock_updates", 6, "clock_updates"),
                                 ("initial_tree", 8, "initial_tree"), ("final_tree", 8, "final_tree")):
            a = arr(name, words)
            setattr(v, fld, a.ctypes.data if a.shape[0] else None)
            setattr(v, "n_" + fld, a.shape[0])
        for i, x in enumerate(arrays.get("roots", [0, 0])):
            v.roots[i] = int(x)
        for i, x in enumerate(arrays.get("ranges", [0] * 6)):
            v.ranges[i] = int(x)
        self._v = v
        self.steps = total

    @property
    def view(self):
        return C.cast(C.pointer(self._v), C.c_void_p)

    def free(self):
        pass


def partial_merkle_tree(cells, initial=True, ranges=(0, 0, 0, 0, 0, 0), lib=None):
    """build_partial_merkle_tree (crates/prover/src/adapter/merkle.rs:183-295) over cells = [(address, v0, v1, v2, v3), ...]:
    (nodes n x 8, root).  Host code."""
    L = lib or load_library()
    c = np.ascontiguousarray(np.array(cells, dtype=np.uint32).reshape(-1, 5))
    cap = max(4096, 64 * c.shape[0] * 31)
    out = np.zeros((cap, 8), dtype=np.uint32)
    n, root = C.c_uint64(0), C.c_uint32(0)
    rc = L.cm_adapter_partial_tree(_p(c), C.c_uint32(c.shape[0]), C.c_int32(1 if initial else 0), (C.c_uint32 * 6)(*ranges), _p(out),
                                   C.c_uint64(cap), C.byref(n), C.byref(root))
    if rc:
        raise _lib_error(L, rc)
    return out[:n.value].copy(), root.value


def _cfg(cfg):
    if cfg is None:
        return None
    return (C.c_uint32 * 4)(*cfg)  # pow_bits, log_blowup_factor, log_last_layer_degree_bound, n_queries


def _backend_prove(self, host_input, cfg=None):
    """prove_cairo_m (crates/prover/src/prover.rs:23): upload + prove."""
    h = C.c_void_p()
    self._ck(self.L.cm_prove_segment(host_input.view, _cfg(cfg), C.byref(h)))
    return Proof(self.L, h)


def _backend_upload(self, host_input):
    h = C.c_void_p()
    self._ck(self.L.cm_input_upload(host_input.view, C.byref(h)))
    return h


def _backend_prove_device(self, dev_input, cfg=None):
    h = C.c_void_p()
    self._ck(self.L.cm_prove_device(dev_input, _cfg(cfg), C.byref(h)))
    return Proof(self.L, h)


Backend.prove = _backend_prove
Backend.upload_input = _backend_upload
Backend.prove_device = _backend_prove_device
Backend.free_input = lambda self, h: self.L.cm_input_free(h)


# ---- runner segments and the device-side adapter (include/cairom_hip.h: cm_runner_segment) -----------------
N_OPCODE_COMPONENTS = 26


class ProverInputView(C.Structure):
    """cm_prover_input (read-only mirror, used to compare adapters field by field)."""
    _fields_ = [("regs", C.c_uint32 * 4),
                ("bundles", C.c_void_p * N_OPCODE_COMPONENTS), ("n_bundles", C.c_uint64 * N_OPCODE_COMPONENTS),
                ("data_accesses", C.c_void_p), ("n_data_accesses", C.c_uint64),
                ("initial_memory", C.c_void_p), ("n_initial_memory", C.c_uint64),
                ("final_memory", C.c_void_p), ("n_final_memory", C.c_uint64),
                ("clock_updates", C.c_void_p), ("n_clock_updates", C.c_uint64),
                ("initial_tree", C.c_void_p), ("n_initial_tree", C.c_uint64),
                ("final_tree", C.c_void_p), ("n_final_tree", C.c_uint64),
                ("roots", C.c_uint32 * 2), ("ranges", C.c_uint32 * 6)]


def prover_input_arrays(view_ptr):
    """cm_prover_input* -> dict of numpy arrays / scalars (copies)."""
    v = C.cast(view_ptr, C.POINTER(ProverInputView)).contents

    def arr(ptr, n, words):
        if not n:
            return np.zeros((0, words), dtype=np.uint32)
        return np.ctypeslib.as_array(C.cast(ptr, _u32p), shape=(int(n), words)).copy()

    out = {"regs": list(v.regs), "roots": list(v.roots), "ranges": list(v.ranges)}
    for i in range(N_OPCODE_COMPONENTS):
        out[f"bundles{i}"] = arr(v.bundles[i], v.n_bundles[i], 12)
    out["data_accesses"] = arr(v.data_accesses, v.n_data_accesses, 4)
    out["initial_memory"] = arr(v.initial_memory, v.n_initial_memory, 7)
    out["final_memory"] = arr(v.final_memory, v.n_final_memory, 7)
    out["clock_updates"] = arr(v.clock_updates, v.n_clock_updates, 6)
    out["initial_tree"] = arr(v.initial_tree, v.n_initial_tree, 8)
    out["final_tree"] = arr(v.final_tree, v.n_final_tree, 8)
    return out


class RunnerSegmentView(C.Structure):
    """cm_runner_segment (read-only mirror)."""
    _fields_ = [("trace", C.c_void_p), ("n_trace", C.c_uint64), ("memory_trace", C.c_void_p), ("n_memory_trace", C.c_uint64),
                ("initial_memory", C.c_void_p), ("n_initial_memory", C.c_uint64), ("ranges", C.c_uint32 * 6),
                ("initial_heap", C.c_void_p), ("n_initial_heap", C.c_uint64)]


def runner_segment_arrays(view_ptr):
    """cm_runner_segment* -> {"trace": (n, 2) (pc, fp), "memory_trace": (n, 5), "initial_memory": (n, 4), "ranges": [6]} (copies)."""
    v = C.cast(view_ptr, C.POINTER(RunnerSegmentView)).contents

    def arr(ptr, n, words):
        if not n:
            return np.zeros((0, words), dtype=np.uint32)
        return np.ctypeslib.as_array(C.cast(ptr, _u32p), shape=(int(n), words)).copy()
    return {"trace": arr(v.trace, v.n_trace, 2), "memory_trace": arr(v.memory_trace, v.n_memory_trace, 5),
            "initial_memory": arr(v.initial_memory, v.n_initial_memory, 4), "ranges": list(v.ranges),
            "initial_heap": arr(v.initial_heap, v.n_initial_heap, 4)}   # index i = the cell at 2^28 - 1 - i


class HostSegment:
    """Raw output of the synthetic VM for one segment (trace, memory log, memory at segment start)."""

    def __init__(self, lib, handle):
        self.L = lib
        self.h = handle
        self.L.cm_host_segment_view.restype = C.c_void_p

    @property
    def view(self):
        return C.c_void_p(self.L.cm_host_segment_view(self.h))

    def free(self):
        if self.h:
            self.L.cm_host_segment_free(self.h)
            self.h = None


def synth_fibonacci_segment(n, max_steps=1 << 30, segment=0, lib=None):
    L = lib or load_library()
    h = C.c_void_p()
    rc = L.cm_synth_fibonacci_segment(C.c_uint32(n), C.c_uint64(max_steps), C.c_uint32(segment), C.byref(h))
    if rc:
        raise _lib_error(L, rc)
    return HostSegment(L, h)


def vm_segment(program, entry_pc=0, args=(), n_returns=0, max_steps=1 << 30, segment=0, lib=None):
    L = lib or load_library()
    words = np.array([w for ins in program for w in ins], dtype=np.uint32)
    lens = np.array([len(ins) for ins in program], dtype=np.uint32)
    a = np.array(list(args), dtype=np.uint32)
    h = C.c_void_p()
    nseg = C.c_uint32(0)
    rc = L.cm_vm_segment(_p(words), _p(lens), C.c_uint32(len(program)), C.c_uint32(entry_pc), _p(a), C.c_uint32(len(a)),
                         C.c_uint32(n_returns), C.c_uint64(max_steps), C.c_uint32(segment), C.byref(h), C.byref(nseg))
    if rc:
        raise _lib_error(L, rc)
    hs = HostSegment(L, h)
    hs.n_segments = nseg.value
    return hs


def _backend_adapt_segment(self, host_segment):
    """import_from_runner_output on the GPU: runner segment -> device-resident ProverInput."""
    h = C.c_void_p()
    self._ck(self.L.cm_adapt_segment_device(host_segment.view, C.byref(h)))
    return h


def _backend_download_input(self, dev_input):
    h = C.c_void_p()
    self._ck(self.L.cm_device_input_download(dev_input, C.byref(h)))
    return HostInput(self.L, h)


Backend.adapt_segment = _backend_adapt_segment
Backend.download_input = _backend_download_input


def _backend_prove_many(self, dev_inputs, inflight=3, cfg=None):
    """Segment pipeline (cm_prove_many): independent segment proofs, up to `inflight` on the GPU at once."""
    n = len(dev_inputs)
    ins = (C.c_void_p * n)(*[d.value if isinstance(d, C.c_void_p) else d for d in dev_inputs])
    outs = (C.c_void_p * n)()
    rc = self.L.cm_prove_many(ins, C.c_uint32(n), _cfg(cfg), C.c_uint32(inflight), outs)
    proofs = [Proof(self.L, C.c_void_p(outs[i])) if outs[i] else None for i in range(n)]
    if rc != 0:
        try:
            self._ck(rc)                       # raises CmError with the first failure's message
        except CmError as e:
            e.partial = proofs                 # the proofs that were built (None where a segment failed): caller frees them
            raise
    return proofs


Backend.prove_many = _backend_prove_many


def _prove_streamed(self, fn, views, inflight, cfg):
    n = len(views)
    ins = (C.c_void_p * n)(*[C.cast(v, C.c_void_p).value for v in views])
    outs = (C.c_void_p * n)()
    rc = fn(ins, C.c_uint32(n), _cfg(cfg), C.c_uint32(inflight), outs)
    proofs = [Proof(self.L, C.c_void_p(outs[i])) if outs[i] else None for i in range(n)]
    if rc != 0:
        try:
            self._ck(rc)
        except CmError as e:
            e.partial = proofs
            raise
    return proofs


def _backend_prove_many_host(self, host_inputs, inflight=3, cfg=None):
    """Streaming ingest (cm_prove_many_host): HOST ProverInputs; input i + 1 uploads while up to `inflight` proofs run."""
    return _prove_streamed(self, self.L.cm_prove_many_host, [h.view for h in host_inputs], inflight, cfg)


def _backend_prove_many_segments(self, host_segments, inflight=3, cfg=None):
    """Streaming ingest from runner segments (cm_prove_many_segments): segment i + 1 goes through the device adapter while up to
    `inflight` proofs run."""
    return _prove_streamed(self, self.L.cm_prove_many_segments, [h.view for h in host_segments], inflight, cfg)


Backend.prove_many_host = _backend_prove_many_host
Backend.prove_many_segments = _backend_prove_many_segments


def _backend_set_preprocessed_cache(self, on):
    """cm_set_preprocessed_cache: keep the committed preprocessed tree (tree 0) between proofs; off by default."""
    self._ck(self.L.cm_set_preprocessed_cache(C.c_int32(1 if on else 0)))


Backend.set_preprocessed_cache = _backend_set_preprocessed_cache
Backend.set_twiddle_cache = lambda self, on: self._ck(self.L.cm_set_twiddle_cache(C.c_int32(1 if on else 0)))
Backend.set_device_tail = lambda self, on: self._ck(self.L.cm_set_device_tail(C.c_int32(1 if on else 0)))
Backend.pool_trim = lambda self: self._ck(self.L.cm_pool_trim())   # this thread's cached device blocks back to the driver


def _backend_mem_info(self):
    """cm_device_mem_info: (free, total) bytes of the library device's HBM."""
    f, t = C.c_uint64(0), C.c_uint64(0)
    self._ck(self.L.cm_device_mem_info(C.byref(f), C.byref(t)))
    return f.value, t.value


Backend.mem_info = _backend_mem_info


# ---- compiled-program JSON (crates/common/src/program.rs:143-170, instruction.rs:609-655) ----------------------
def load_program_json(text):
    """Compiled `Program` as the reference serialises it with serde_json: {"data": [{"Instruction": ["0x9", "0x1", ...]}
    | {"Value": [[a, b], [c, d]]}, ...], "entrypoints": {name: {"pc": n, "params": [...], "returns": [...]}},
    "metadata": {...}}.  Returns (cells, entrypoints): `cells` = one word list per program datum — instruction words
    (opcode first, 1..6 words) or the 4 words of a raw QM31 value — in the form cm_vm_run / vm_run take."""
    import json
    doc = json.loads(text)
    unknown = set(doc) - {"data", "entrypoints", "metadata"}
    if unknown:
        raise ValueError(f"unknown Program fields {sorted(unknown)} (the reference denies unknown fields)")
    cells = []
    for item in doc["data"]:
        if "Instruction" in item:
            words = [int(s, 16) for s in item["Instruction"]]
            if not 1 <= len(words) <= 6:
                raise ValueError("instruction must have 1..6 M31 words")
        elif "Value" in item:
            (a, b), (c, d) = item["Value"]
            words = [int(a), int(b), int(c), int(d)]
        else:
            raise ValueError(f"unknown ProgramData variant {list(item)}")
        if any(w >= 2**31 - 1 for w in words):
            raise ValueError("program word is not a canonical M31")
        cells.append(words)
    entry = {name: {"pc": int(e["pc"]), "n_params": sum(_abi_slots(p["ty"]) for p in e.get("params", [])),
                    "n_returns": sum(_abi_slots(r["ty"]) for r in e.get("returns", []))}
             for name, e in doc.get("entrypoints", {}).items()}
    return cells, entry


def _abi_slots(ty):
    """AbiType::size_in_slots (program.rs:30-42); serde externally-tagged enum: "Felt" | {"Pointer": {...}} | ..."""
    if isinstance(ty, str):
        return {"Felt": 1, "Bool": 1, "U32": 2, "Unit": 0}[ty]
    (tag, body), = ty.items()
    if tag == "Pointer":
        return 1
    if tag == "Tuple":
        return sum(_abi_slots(t) for t in body)
    if tag == "Struct":
        return sum(_abi_slots(t) for _, t in body["fields"])
    if tag == "FixedSizeArray":
        return int(body["size"]) * _abi_slots(body["element"])
    raise ValueError(f"unknown AbiType {tag}")


def program_to_json(cells, entrypoints=None):
    """Inverse of load_program_json for instruction-only programs (hex strings like `format!("0x{:x}")`)."""
    import json
    return json.dumps({"data": [{"Instruction": [f"0x{w:x}" for w in ins]} for ins in cells],
                       "entrypoints": entrypoints or {}, "metadata": {}})


# ---- per-component AIR ops (include/cairom_hip.h) -----------------------------------------------------
N_COMPONENTS = 34
N_PREPROCESSED = 7
PREPROCESSED_LOG = (18, 18, 18, 18, 8, 16, 20)
RELATION_WORDS = 8 * 4 + 8 * 16 * 4   # cm_relations: z[8][4], alpha_pow[8][16][4]


def _b_component_info(self, cid):
    a, b, c = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
    self._ck(self.L.cm_component_info(C.c_int32(cid), C.byref(a), C.byref(b), C.byref(c)))
    return a.value, b.value, c.value


def _b_component_log_size(self, dev_input, cid):
    lg = C.c_uint32(0)
    self._ck(self.L.cm_component_log_size(dev_input, C.c_int32(cid), C.byref(lg)))
    return lg.value


def _b_trace_write(self, dev_input, cid, cols):
    self._ck(self.L.cm_trace_write(dev_input, C.c_int32(cid), self._harr(cols), C.c_uint64(0)))


def _b_histogram(self, cid, trace_cols, log_size, rc8, rc16, rc20, bitwise):
    self._ck(self.L.cm_histogram(C.c_int32(cid), self._harr(trace_cols), C.c_uint32(log_size), C.c_uint64(rc8), C.c_uint64(rc16),
                                 C.c_uint64(rc20), C.c_uint64(bitwise), C.c_uint64(0)))


def _b_preprocessed_column(self, pp_id, col):
    self._ck(self.L.cm_preprocessed_column(C.c_int32(pp_id), C.c_uint64(col), C.c_uint64(0)))


def _b_interaction_write(self, cid, trace_cols, preprocessed, log_size, rel_words, out_cols):
    r = np.ascontiguousarray(rel_words, dtype=np.uint32)
    assert r.size == RELATION_WORDS
    cs = np.zeros(4, dtype=np.uint32)
    self._ck(self.L.cm_interaction_write(C.c_int32(cid), self._harr(trace_cols), self._harr(preprocessed), C.c_uint32(log_size),
                                         _p(r), self._harr(out_cols), _p(cs), C.c_uint64(0)))
    return cs


def _b_constraints_accumulate(self, cid, trace_lde, interaction_lde, preprocessed_lde, log_size, rel_words, coeff_words,
                              claimed_sum, acc4):
    r = np.ascontiguousarray(rel_words, dtype=np.uint32)
    co = np.ascontiguousarray(coeff_words, dtype=np.uint32)
    cs = np.ascontiguousarray(claimed_sum, dtype=np.uint32)
    self._ck(self.L.cm_constraints_accumulate(C.c_int32(cid), self._harr(trace_lde), self._harr(interaction_lde),
                                              self._harr(preprocessed_lde), C.c_uint32(log_size), _p(r), _p(co), _p(cs),
                                              self._harr(acc4), C.c_uint64(0)))


def _b_fri_decompose(self, f4, log_n):
    lam = np.zeros(4, dtype=np.uint32)
    self._ck(self.L.cm_fri_decompose(self._harr(f4), C.c_uint32(log_n), _p(lam), C.c_uint64(0)))
    return lam


Backend.component_info = _b_component_info
Backend.component_log_size = _b_component_log_size
Backend.trace_write = _b_trace_write
Backend.histogram = _b_histogram
Backend.preprocessed_column = _b_preprocessed_column
Backend.interaction_write = _b_interaction_write
Backend.constraints_accumulate = _b_constraints_accumulate
Backend.fri_decompose = _b_fri_decompose


def _b_accumulate(self, dst4, src4, n):
    self._ck(self.L.cm_accumulate(self._harr(dst4), self._harr(src4), C.c_uint64(n), C.c_uint64(0)))


def _b_secure_powers(self, felt, n):
    f = np.ascontiguousarray(felt, dtype=np.uint32)
    out = np.zeros(4 * n, dtype=np.uint32)
    self._ck(self.L.cm_generate_secure_powers(_p(f), C.c_uint64(n), _p(out)))
    return out.reshape(n, 4)


def _b_col_zero(self, h, n):
    self._ck(self.L.cm_col_zero(C.c_uint64(h), C.c_uint64(n), C.c_uint64(0)))


Backend.accumulate = _b_accumulate
Backend.secure_powers = _b_secure_powers
Backend.col_zero = _b_col_zero
