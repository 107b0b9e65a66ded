#!/bin/bash
# Round-end measurement set on the GPU box (writes under gpurun_out/, copy what is to be judged into profiles/):
#   tools/measure_round.sh <tag>     e.g. r01f
# 1. default bench (the line the driver records)            -> gpurun_out/<tag>_bench_n1.json
# 2. rocprofv3 --kernel-trace --stats of the same command     -> gpurun_out/<tag>_stats (rocpd db) + summary csv
# 3. two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs)   -> gpurun_out/<tag>_pmc_traffic.json
# 4. one PMC pass of SQ issue / stall counters                 -> gpurun_out/<tag>_pmc_sq.json
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python bench.py > gpurun_out/${tag}_bench_n1.json 2> gpurun_out/${tag}_bench_n1.err
# the same line with the 2^26-row leg (BASELINE configs[4] on one GPU: ~116 GiB, about two more minutes): the larger-sizes model
python bench.py --no-cpu-baseline --no-end-to-end --alt-fib-n 0 --pipelined 0 --cached-setup-steps 0 --sharded-one-rank-blocks 0 --big-mixed-iters 1545000 > gpurun_out/${tag}_bench_big_legs.json 2> gpurun_out/${tag}_bench_big_legs.err
# GPU idle inside a proof (kernel + copy timeline of the same workload)
tools/gaps.sh ${tag} --list > /dev/null 2>&1
BENCH="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --pipelined 0 --no-end-to-end --alt-fib-n 0 --big-fib-n 0 --cached-setup-steps 0 --sharded-one-rank-blocks 0"
rm -rf gpurun_out/${tag}_stats gpurun_out/${tag}_pmc_fetch gpurun_out/${tag}_pmc_write
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_stats -o s -- $BENCH > gpurun_out/${tag}_bench_under_rocprof.json 2> gpurun_out/${tag}_stats.err
db=$(ls gpurun_out/${tag}_stats/*/*results.db gpurun_out/${tag}_stats/*results.db 2>/dev/null | head -1)
python tools/rocpd_summary.py "$db" --csv gpurun_out/${tag}_rocprofv3_kernel_stats.csv \
  --header "rocprofv3 --kernel-trace --stats -- $BENCH (MI355X; 5 proofs of fibonacci_loop n=419000 in the trace: 1 warmup + 1 fully instrumented + 2 timed + 1 verified)" | tail -3
P1="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --pipelined 0 --no-kprof --no-end-to-end --alt-fib-n 0 --big-fib-n 0 --cached-setup-steps 0 --sharded-one-rank-blocks 0"   # 3 proofs: warmup, timed, verified
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/${tag}_pmc_fetch -o f -- $P1 > /dev/null 2> gpurun_out/${tag}_pmc_fetch.err
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/${tag}_pmc_write -o w -- $P1 > /dev/null 2> gpurun_out/${tag}_pmc_write.err
fdb=$(ls gpurun_out/${tag}_pmc_fetch/*/*results.db gpurun_out/${tag}_pmc_fetch/*results.db 2>/dev/null | head -1)
wdb=$(ls gpurun_out/${tag}_pmc_write/*/*results.db gpurun_out/${tag}_pmc_write/*results.db 2>/dev/null | head -1)
python tools/pmc_traffic.py "$fdb" "$wdb" --proofs 3 --json gpurun_out/${tag}_pmc_traffic.json | tail -15
# third PMC pass: SQ issue / stall counters (8 SQ slots per pass) -> is a class issue-bound or parked on memory?
rm -rf gpurun_out/${tag}_pmc_sq
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY \
  -d gpurun_out/${tag}_pmc_sq -o q -- $P1 > /dev/null 2> gpurun_out/${tag}_pmc_sq.err
qdb=$(ls gpurun_out/${tag}_pmc_sq/*/*results.db gpurun_out/${tag}_pmc_sq/*results.db 2>/dev/null | head -1)
(cd tools && python pmc_sq.py "../$qdb" --proofs 3 --json ../gpurun_out/${tag}_pmc_sq.json) | tail -20
rm -rf gpurun_out/${tag}_stats gpurun_out/${tag}_pmc_fetch gpurun_out/${tag}_pmc_write gpurun_out/${tag}_pmc_sq   # the databases are large
