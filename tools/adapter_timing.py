"""Timing of the device-side adapter vs the host adapter on the metric workload (development tool)."""
import sys, time
sys.path.insert(0, '.')
from cairo_m_amd import Backend
from cairo_m_amd.lib import synth_fibonacci, synth_fibonacci_segment
be = Backend(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 419_000
t = time.perf_counter(); hs = synth_fibonacci_segment(n); t_vm = time.perf_counter() - t
t = time.perf_counter(); hi = synth_fibonacci(n); t_vm_adapt = time.perf_counter() - t
print(f"synthetic VM only {t_vm*1e3:.1f} ms; VM + host adapter {t_vm_adapt*1e3:.1f} ms -> host adapter ~{(t_vm_adapt-t_vm)*1e3:.1f} ms")
for i in range(3):
    t = time.perf_counter(); dev = be.adapt_segment(hs); dt = time.perf_counter() - t
    print(f"device adapter (incl. upload of the runner output) {dt*1e3:.1f} ms")
    be.free_input(dev)
t = time.perf_counter(); dev = be.upload_input(hi); print(f"upload of a host-adapted ProverInput {1e3*(time.perf_counter()-t):.1f} ms")
