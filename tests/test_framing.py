"""Named framing switches (include/cairom_hip.h cm_set_framing, oracle/oframing.hpp): the Stwo-side conventions that no
in-tree reference vector settles each have their alternate written out on BOTH sides.  CPU half: parsing, and the oracle
prover / oracle verifier / product verifier (host code) agree with each other under the all-alternate setting and refuse a
proof made under a different setting.  The GPU half (tests/test_gpu_framing.py) compares HIP with the oracle under every
setting."""
import ctypes as C

import numpy as np
import pytest

from cairo_m_amd.lib import CmError, get_framing, load_library, set_framing
from tests.ref_inputs import unchanged_memory_input

ALT = "mix_u64=u32s,hash_node=rfc,sample_batch=sorted,pcs_mix=blq"
DEFAULT = "mix_u64=raw,hash_node=raw,sample_batch=insertion,pcs_mix=bql"


def _product_verify(words, L):
    w = np.ascontiguousarray(words, dtype=np.uint32)
    return L.cm_verify_proof_words(w.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_uint64(w.size), None)


def test_framing_spec_parsing():
    set_framing("")
    assert get_framing() == DEFAULT
    set_framing("hash_node=rfc")
    assert get_framing() == DEFAULT.replace("hash_node=raw", "hash_node=rfc")
    set_framing(" mix_u64 = u32s , pcs_mix=blq".replace(" = ", "="))
    assert "mix_u64=u32s" in get_framing() and "pcs_mix=blq" in get_framing() and "hash_node=raw" in get_framing()
    for bad in ("hash_node=md5", "nonsense=1", "hash_node"):
        with pytest.raises(CmError):
            set_framing(bad)
    assert "mix_u64=u32s" in get_framing()          # a rejected spec leaves the setting alone
    set_framing("default")
    assert get_framing() == DEFAULT


def test_alternate_framing_is_consistent_across_oracle_and_product_verifier(oracle):
    """The hand-built input of the reference's test_prove_and_verify_unchanged_memory (crates/prover/tests/prover.rs:33-112)
    proved by the oracle under the all-alternate framing: both verifiers accept under that setting, both refuse under the
    default one (and a default-framing proof is refused under the alternate setting)."""
    L = load_library()
    inp = unchanged_memory_input()
    try:
        oracle.set_framing(ALT)
        set_framing(ALT, L)
        w_alt, _, tr_alt = oracle.prove(inp.view, transcript=True)
        assert oracle.verify(w_alt)[0] == 0
        assert _product_verify(w_alt, L) == 0
        oracle.set_framing("")
        set_framing("", L)
        assert oracle.verify(w_alt)[0] != 0
        assert _product_verify(w_alt, L) != 0
    finally:
        oracle.set_framing("")
        set_framing("", L)
    # transcript: the first four entries are PcsConfig::mix_into — pow_bits, log_blowup, then (alt) last-layer bound, n_queries
    assert [e["op"] for e in tr_alt[:4]] == ["mix_u64"] * 4
    assert [e["words"][0] for e in tr_alt[:4]] == [16, 1, 0, 80]


def test_malformed_env_framing_is_a_hard_error():
    """CM_FRAMING is read once, on first use; a value that does not parse must not silently become the default framing (a proof
    made under a framing the operator did not ask for is a wrong proof): cm_get_framing reports -1, cm_set_framing repairs it."""
    import subprocess
    import sys
    code = (
        "import ctypes as C, sys\n"
        "from cairo_m_amd.lib import load_library\n"
        "L = load_library()\n"
        "buf = C.create_string_buffer(256)\n"
        "rc = L.cm_get_framing(buf, C.c_size_t(256))\n"
        "assert rc == -1, rc\n"
        "assert L.cm_set_framing(b'hash_node=rfc') == 0\n"
        "rc = L.cm_get_framing(buf, C.c_size_t(256))\n"
        "assert rc > 0 and b'hash_node=rfc' in buf.value, (rc, buf.value)\n"
        "print('ok')\n")
    import os
    env = dict(os.environ, CM_FRAMING="mix_u64=sometimes", PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
