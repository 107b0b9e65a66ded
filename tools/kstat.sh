#!/bin/bash
# per-kernel average durations of a short bench run: tools/kstat.sh <tag> <grep pattern>
tag=$1; pat=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/${tag}_ks
rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_ks -o s -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --pipelined 0 --no-end-to-end --no-kprof > /dev/null 2> gpurun_out/${tag}_ks.err
db=$(ls gpurun_out/${tag}_ks/*/*results.db gpurun_out/${tag}_ks/*results.db 2>/dev/null | head -1)
python tools/rocpd_summary.py "$db" --csv gpurun_out/${tag}_kstat.csv --header "kstat" > /dev/null
grep -E "$pat" gpurun_out/${tag}_kstat.csv
rm -rf gpurun_out/${tag}_ks
