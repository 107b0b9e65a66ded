#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_sharded.py -x -q -m gpu > gpurun_out/r06i_sharded_tests.txt 2>&1; tail -3 gpurun_out/r06i_sharded_tests.txt
grep -q failed gpurun_out/r06i_sharded_tests.txt && exit 1
tools/r05_sharded1.sh > gpurun_out/r06i_sharded_world1.txt 2>&1; grep -v Librccl gpurun_out/r06i_sharded_world1.txt
