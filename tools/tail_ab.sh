# development: FRI-tail cutoff sweep (CM_FRI_TAIL_LOG) on the two bench sizes
for r in 1 2; do for v in ${TAIL_LOGS:-13 12 11 10 9}; do
for n in 100000 419000; do
CM_FRI_TAIL_LOG=$v python bench.py --steps 8 --warmup 2 --no-cpu-baseline --pipelined 0 --fib-n $n 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['roofline']['kernels'];print('tail_log $v n=$n', round(d['ms_per_step'],3), 'fri_commit', round(d['phase_ms']['fri_commit'],3), 'k_fri_tail', round(k['k_fri_tail']['ms_per_step'],3))"
done; done; done
