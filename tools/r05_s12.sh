#!/bin/bash
# round 5, session 12: the flag self-test — plain run keeps the flags, a rocprofv3 --pmc run falls back to events; PMC traffic passes
tag=r05q
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 200 python tools/lone_loop.py 2>&1 | tail -3
P1="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --pipelined 0 --no-kprof --no-end-to-end --alt-fib-n 0 --big-fib-n 0"
rm -rf gpurun_out/${tag}_pmc_fetch gpurun_out/${tag}_pmc_write gpurun_out/${tag}_pmc_sq
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/${tag}_pmc_fetch -o f -- $P1 > /dev/null 2> gpurun_out/${tag}_pmc_fetch.err; echo "fetch pass rc $?"; grep -m2 "cairom_hip\]" gpurun_out/${tag}_pmc_fetch.err
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/${tag}_pmc_write -o w -- $P1 > /dev/null 2> gpurun_out/${tag}_pmc_write.err; echo "write pass rc $?"
fdb=$(ls gpurun_out/${tag}_pmc_fetch/*/*results.db gpurun_out/${tag}_pmc_fetch/*results.db 2>/dev/null | head -1)
wdb=$(ls gpurun_out/${tag}_pmc_write/*/*results.db gpurun_out/${tag}_pmc_write/*results.db 2>/dev/null | head -1)
python tools/pmc_traffic.py "$fdb" "$wdb" --proofs 3 --json gpurun_out/${tag}_pmc_traffic.json | tail -15
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY \
  -d gpurun_out/${tag}_pmc_sq -o q -- $P1 > /dev/null 2> gpurun_out/${tag}_pmc_sq.err; echo "sq pass rc $?"
qdb=$(ls gpurun_out/${tag}_pmc_sq/*/*results.db gpurun_out/${tag}_pmc_sq/*results.db 2>/dev/null | head -1)
(cd tools && python pmc_sq.py "../$qdb" --proofs 3 --json ../gpurun_out/${tag}_pmc_sq.json) | tail -12
rm -rf gpurun_out/${tag}_pmc_fetch gpurun_out/${tag}_pmc_write gpurun_out/${tag}_pmc_sq
