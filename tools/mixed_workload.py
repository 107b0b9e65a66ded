"""Mixed u32/felt loop at scale (stand-in for BASELINE configs[2]/[4]): VM -> device adapter -> HIP prover -> both
verifiers.  Development tool: prints timings and the per-phase split."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from cairo_m_amd import Backend
from cairo_m_amd.lib import vm_segment
from tests.oracle_binding import Oracle
from tests.test_oracle_air import u32_loop_program
be = Backend(0)
orc = Oracle('oracle/liboracle.so')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 95_000
t = time.perf_counter(); hs = vm_segment(u32_loop_program(n), entry_pc=0, args=(), n_returns=0); print('vm', round(time.perf_counter() - t, 3), 's')
t = time.perf_counter(); dev = be.adapt_segment(hs); print('device adapter', round((time.perf_counter() - t) * 1e3, 1), 'ms')
for i in range(3):
    t = time.perf_counter(); p = be.prove_device(dev); dt = time.perf_counter() - t
    st = p.stats()
    print('prove ms', round(dt * 1e3, 2), 'steps', st['steps'], 'cells', st['cells'], 'cells/s %.3e' % (st['cells'] / dt))
    if i == 2:
        print({k: round(v, 2) for k, v in st['phase_ms'].items()})
        print('product verifier', p.verify(), 'oracle verifier', orc.verify(p.words()))
    p.free()
