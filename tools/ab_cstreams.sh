for i in 1 2; do
for v in "0,1,2,3,7,4,5,6" "0,2,4,6,1,3,5,7" "0,1,4,5,2,3,6,7" "1,3,5,7,0,2,4,6" "0,4,1,5,2,6,3,7"; do
  export CM_CSTREAMS=$v
  echo -n "$v "
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline --pipelined 0 --no-end-to-end --alt-fib-n 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['phase_ms']['constraints'],3), round(d['phase_ms']['quotients'],3))"
done; done
