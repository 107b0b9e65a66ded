#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_poly_merkle.py tests/test_gpu_prove.py tests/test_gpu_fri_quotients.py -x -q -m gpu > gpurun_out/r05x_tests.txt 2>&1; tail -3 gpurun_out/r05x_tests.txt
grep -q passed gpurun_out/r05x_tests.txt || exit 1
timeout 900 python tools/ab_switch.py --reps 12 merkle_top_npb64 oods_poll merkle_top_npb64 > gpurun_out/r05x_ab_switch.txt 2>&1
cat gpurun_out/r05x_ab_switch.txt
tools/gaps.sh r05x --list > /dev/null; grep "k_merkle_top" gpurun_out/r05x_gaps.txt | awk '{print $6, $10}' | sort -n | tr '\n' ';'
