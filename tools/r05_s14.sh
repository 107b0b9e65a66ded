#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_adapter.py tests/test_gpu_casm_programs.py -x -q -m gpu > gpurun_out/r05t_tests.txt 2>&1; tail -5 gpurun_out/r05t_tests.txt
