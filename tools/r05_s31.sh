#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_prove.py -x -q -m gpu > gpurun_out/r06n_tests.txt 2>&1; tail -2 gpurun_out/r06n_tests.txt
grep -q failed gpurun_out/r06n_tests.txt && exit 1
timeout 900 python tools/ab_switch.py --reps 16 tail_flags defer_teardown tail_flags > gpurun_out/r06n_ab_switch.txt 2>&1
cat gpurun_out/r06n_ab_switch.txt
tools/gaps.sh r06n --list > /dev/null; grep "k_tail\|k_fri_tail" gpurun_out/r06n_gaps.txt | tail -6
