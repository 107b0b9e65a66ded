#!/bin/bash
# On the GPU box: the GPU test suite (or "$@") against the host-ASAN build of the library.  Output -> gpurun_out/<tag>_asan.txt
#   tools/asan_run.sh r04 [pytest args...]
tag=${1:-rXX}; shift
RT=$(ls /usr/lib/x86_64-linux-gnu/libasan.so.* | head -1)   # the system runtime, NOT ROCm's libclang_rt.asan (tools/asan_build.sh)
export CAIROM_HIP_LIB=$PWD/cairo_m_amd/libcairom_hip_asan.so
# detect_leaks=0: CPython and the HIP runtime keep process-lifetime allocations; protect_shadow_gap=0: the ROCm runtime maps
# fixed addresses inside ASAN's shadow gap
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:halt_on_error=0:log_path=gpurun_out/${tag}_asan_report
ARGS=("$@"); [ ${#ARGS[@]} -eq 0 ] && # (not under the sanitizer: the tests that start torch in a child process — bench.py, the sharded prover's torch.distributed ranks:
# torch's own dlopen / exception paths fail under a preloaded sanitizer runtime ("Error in dlopen: libcaffe2_nvrtc.so"), before
# anything of this library runs — and the 116 GiB workload)
ARGS=(tests -m gpu -q --ignore=tests/test_gpu_bench_contract.py --ignore=tests/test_gpu_sharded.py --deselect tests/test_gpu_workloads.py::test_configs4_all_opcodes_at_2pow26_rows_verifies)
# libstdc++ is preloaded as well: python loads it late (dlopen), and the runtime resolves the real __cxa_throw when IT initialises
# — without it every C++ exception (the library's own error statuses included) dies in the interceptor
LD_PRELOAD="$RT /usr/lib/x86_64-linux-gnu/libstdc++.so.6" python -m pytest "${ARGS[@]}" > gpurun_out/${tag}_asan.txt 2>&1
tail -5 gpurun_out/${tag}_asan.txt
ls gpurun_out/${tag}_asan_report* 2>/dev/null | head
