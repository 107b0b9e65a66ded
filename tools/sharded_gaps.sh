#!/bin/bash
# GPU idle time inside one SHARDED proof on one rank (in-library RCCL): tools/sharded_gaps.sh <tag> [--list] -> gpurun_out/<tag>_sharded_gaps.txt
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/${tag}_stl
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/${tag}_stl -- \
  python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29517 -m cairo_m_amd.sharded \
  --fib-n 419000 --comm rccl --steps 4 > gpurun_out/${tag}_stl.out 2> gpurun_out/${tag}_stl.err
k=$(ls -S gpurun_out/${tag}_stl/*/*_kernel_trace.csv | head -1); m=${k%_kernel_trace.csv}_memory_copy_trace.csv
python tools/timeline_gaps.py $k $m "$@" > gpurun_out/${tag}_sharded_gaps.txt
head -${GAPS_HEAD:-30} gpurun_out/${tag}_sharded_gaps.txt
tail -2 gpurun_out/${tag}_stl.out
rm -rf gpurun_out/${tag}_stl
