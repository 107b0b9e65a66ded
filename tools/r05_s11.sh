#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_sharded.py -x -q -m gpu > gpurun_out/r05m_sharded_tests.txt 2>&1; tail -4 gpurun_out/r05m_sharded_tests.txt
tools/r05_sharded1.sh 2>&1 | grep -v "^Librccl\|amdgpu.ids" | tee gpurun_out/r05m_sharded_world1.txt
