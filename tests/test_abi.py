"""The C-ABI library loads on a CPU-only box and exports every symbol include/cairom_hip.h declares;
compute entry points fail loudly (no CPU fallback) when no GPU is present."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "cairom_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cm_[a-z0-9_]+)\s*\(", text)))


def test_all_declared_symbols_exported():
    from cairo_m_amd.lib import load_library
    L = load_library()
    syms = declared_symbols()
    assert len(syms) >= 40
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cairo_m_amd import Backend, CmError
    with pytest.raises(CmError):
        Backend(0)


def test_process_wide_switches_reject_what_they_do_not_know():
    """cm_set_cpu_affinity: 0 never / 1 scoped (default) / 2 sticky; cm_set_tuning: the measurement switches by name (host code:
    no GPU needed to set them)"""
    from cairo_m_amd.lib import load_library
    L = load_library()
    before = L.cm_get_cpu_affinity()
    try:
        for mode in (0, 2, 1):
            assert L.cm_set_cpu_affinity(C.c_int32(mode)) == 0 and L.cm_get_cpu_affinity() == mode
        assert L.cm_set_cpu_affinity(C.c_int32(3)) != 0 and L.cm_get_cpu_affinity() == 1
    finally:
        L.cm_set_cpu_affinity(C.c_int32(before))
    for key in (b"oods_poll", b"oods_host_write", b"stage_copy_kernel", b"stage_lazy_events", b"defer_teardown", b"flag_join", b"flag_fork",
                b"commit_prep_early", b"trace_hist_fuse", b"logup_defer"):
        assert L.cm_set_tuning(key, C.c_int32(0)) == 0 and L.cm_set_tuning(key, C.c_int32(1)) == 0
    assert L.cm_set_tuning(b"fri_top_fuse", C.c_int32(1)) == 0 and L.cm_set_tuning(b"fri_top_fuse", C.c_int32(0)) == 0
    assert L.cm_set_tuning(b"oods_split", C.c_int32(650)) == 0 and L.cm_set_tuning(b"oods_split", C.c_int32(780)) == 0
    assert L.cm_set_tuning(b"oods_split", C.c_int32(1001)) != 0
    assert L.cm_set_tuning(b"no_such_switch", C.c_int32(1)) != 0
    assert L.cm_set_tuning(None, C.c_int32(1)) != 0
