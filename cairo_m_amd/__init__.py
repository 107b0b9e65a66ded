"""cairo_m_amd — MI355X (gfx950) proving backend for the Cairo-M zkVM.

Python is glue only: this package loads ``libcairom_hip.so`` (hand-written HIP kernels behind a
C ABI, ``include/cairom_hip.h``) through ctypes and mirrors the reference's prover entry points
(``prove_cairo_m`` — /root/reference/crates/prover/src/prover.rs:23).  There is no CPU fallback:
every compute call fails loudly when the library or a GPU is missing.
"""
from .lib import Backend, load_library, LIB_PATH, CmError  # noqa: F401
