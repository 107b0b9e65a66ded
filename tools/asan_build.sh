#!/bin/bash
# Host-side AddressSanitizer build of the library next to the shipped one (SURVEY §5 "race detection / sanitizers"):
#   tools/asan_build.sh            -> cairo_m_amd/libcairom_hip_asan.so  (objects in cairo_m_amd/csrc/build_asan)
# Device code is NOT instrumented (-fno-gpu-sanitize: the image has no xnack+ ASAN device runtime); what this build checks is the
# host side of the prover — the decommitment sections and gather plans, the upload ring, the pinned landing buffers, proof
# assembly, the C ABI marshalling — under the real GPU workload.  Run on the GPU box with tools/asan_run.sh.
set -e
cd "$(dirname "$0")/../cairo_m_amd/csrc"
# Compile with the instrumentation, LINK WITHOUT a sanitizer runtime: ROCm's own libclang_rt.asan intercepts
# hsa_amd_memory_pool_allocate for device-side ASAN and aborts inside hipInit on a stock (non-ASAN) ROCm stack; the __asan_*
# symbols are resolved at load time from the system's libasan (gcc's libasan.so.6 has all 44 of them), which tools/asan_run.sh
# preloads.
make -j8 BUILD=build_asan TARGET=../libcairom_hip_asan.so \
  EXTRA="-fsanitize=address -fno-gpu-sanitize -g -fno-omit-frame-pointer"
