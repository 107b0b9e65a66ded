#!/bin/bash
# One rank = the whole proof through the sharded path (the "single-rank tax" of cm_prove_sharded), in-library RCCL communicator:
#   tools/sharded_world1.sh <tag> [env assignments ...]   -> gpurun_out/<tag>_sharded_world1.txt
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for e in "$@"; do export "$e"; done
python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29517 -m cairo_m_amd.sharded \
  --fib-n 419000 --comm rccl --steps ${STEPS:-12} --check-single 2>&1 | grep -v Warning | tail -3 > gpurun_out/${tag}_sharded_world1.txt
cat gpurun_out/${tag}_sharded_world1.txt
