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
