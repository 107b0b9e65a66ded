"""CPU twin of tests/test_gpu_logup_golden.py: the ORACLE's own LogUp generator (oracle/ologup.hpp, written independently of the
product's LogupStream) against the reference-derived LogUp vectors (tests/golden/air_logup_vectors.npz, made by
tools/rsref/rs_logup.py from the reference's `write_interaction_trace` text).  Same program as the golden run
(all_opcodes_program(iters, seed) on the library's VM + adapter), same relation parameters; every column but the last equals
the per-row running sum of the fractions, the last one is the row-order-free running sum (see the GPU test), the claimed sum
is the sum of the row totals."""
import os

import numpy as np
import pytest

from cairo_m_amd.lib import vm_run
from cairo_m_amd.workloads import all_opcodes_program

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOG = np.load(os.path.join(ROOT, "tests", "golden", "air_logup_vectors.npz"))
from tests.test_air_witness_golden import OPCODE_FILES  # noqa: E402
from tests.test_gpu_logup_golden import NAMES, P, relation_words  # noqa: E402  (plain-Python helpers; nothing there touches a GPU at import)


@pytest.fixture(scope="module")
def golden_input():
    prog, steps = all_opcodes_program(int(LOG["iters"][0]), int(LOG["seed"][0]))
    inp = vm_run(prog, entry_pc=0, args=(), n_returns=0)
    assert inp.steps == steps
    yield inp
    inp.free()


# (clock_update, id 28, is left to the GPU test: its golden rows are synthetic entries, not this program's — it has none)
@pytest.mark.parametrize("cid", range(28), ids=NAMES[:28])
def test_oracle_logup_columns_equal_reference_derived_fractions(oracle, golden_input, cid):
    name = NAMES[cid]
    want = LOG[name].astype(np.int64)
    n_cols, n = want.shape[0], want.shape[1]
    log = n.bit_length() - 1
    cols, cs = oracle.component_interaction(golden_input.view, cid, relation_words(), 4 * n_cols, log)
    got = cols.astype(np.int64).reshape(n_cols, 4, n)
    last = n_cols - 1
    for j in range(last):
        bad = np.argwhere(got[j].T != want[j])
        assert bad.size == 0, f"{name}: LogUp column {j}: first differing (row, coordinate) {bad[:4].tolist()}"
    total = want[last]
    claimed = total.sum(axis=0) % P
    assert [int(x) for x in cs] == [int(x) for x in claimed], f"{name}: claimed sum"
    shift = claimed * pow(n, P - 2, P) % P
    c = got[last].T
    key = lambda a: sorted(map(tuple, a.tolist()))
    assert key((c - total + shift) % P) == key(c), f"{name}: last column"
