#!/bin/bash
# kernel split of the device adapter (run on the GPU box): tools/adapter_prof.sh [out.csv]
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
out=${1:-gpurun_out/adapter_kernel_stats.csv}
timeout 200 python tools/adapter_prof.py 2>&1 | tail -1
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/adprof -o ad -- python tools/adapter_prof.py --reps 5 > /dev/null 2>&1
db=$(find gpurun_out/adprof -name "*.db" | head -1)
python tools/rocpd_summary.py $db --csv $out --header "rocprofv3 --kernel-trace --stats -- python tools/adapter_prof.py --reps 5 (cm_adapt_segment_device, fibonacci_loop n=419000: 5 calls)" | tail -1
python - "$out" <<PY
import csv,sys
for r in csv.reader(l for l in open(sys.argv[1]) if not l.startswith("#")):
    if r and r[0]!="name": print(r[0].split("<")[0][-40:].ljust(42), r[1].rjust(4), r[2].rjust(10), r[3].rjust(9))
PY
rm -rf gpurun_out/adprof
