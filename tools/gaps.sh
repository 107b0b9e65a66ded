#!/bin/bash
# GPU idle time inside one proof: tools/gaps.sh <tag> [--list]   (env switches are inherited) -> gpurun_out/<tag>_gaps.txt
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/${tag}_tl
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/${tag}_tl -- python bench.py --steps 3 --warmup 2 --pipelined 0 --no-kprof --no-cpu-baseline --no-end-to-end --alt-fib-n 0 --big-fib-n 0 --cached-setup-steps 0 --sharded-one-rank-blocks 0 > /dev/null 2> gpurun_out/${tag}_tl.err
k=$(ls gpurun_out/${tag}_tl/*/*_kernel_trace.csv | head -1); m=$(ls gpurun_out/${tag}_tl/*/*_memory_copy_trace.csv | head -1)
python tools/timeline_gaps.py $k $m "$@" > gpurun_out/${tag}_gaps.txt
head -${GAPS_HEAD:-14} gpurun_out/${tag}_gaps.txt
rm -rf gpurun_out/${tag}_tl
