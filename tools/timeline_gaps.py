#!/usr/bin/env python3
"""GPU-idle gaps of the LAST proof in a rocprofv3 kernel + memory-copy trace (csv output):
   rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -- python bench.py --steps 3 --warmup 2 --pipelined 0 --no-kprof
   tools/timeline_gaps.py DIR/<host>/<pid>_kernel_trace.csv DIR/<host>/<pid>_memory_copy_trace.csv [--list]
Prints span / busy / idle and the largest gaps with the events around them (development tool)."""
import csv, re, sys
k, m = sys.argv[1], sys.argv[2]
ev = []
for r in csv.DictReader(open(k)):
    # (anonymous-namespace kernels — k_join_flag / k_join_collect of pool.hip — keep their names: stripping at the first "(" used to
    # leave them empty, and the listing showed them as unnamed work)
    n = re.sub(r'\(.*', '', r['Kernel_Name'].replace('(anonymous namespace)::', '')).replace('cm::', '').replace('void ', '')[:44]
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), n, r['Queue_Id'], r['Grid_Size_X']))
for r in csv.DictReader(open(m)):
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r['Direction'][12:], '-', ''))
ev.sort()
gi = [i for i, e in enumerate(ev) if 'k_tail_gather' in e[2]]   # device-side tail (round 5): the last kernel of a proof
if not gi:
    gi = [i for i, e in enumerate(ev) if 'k_grind' in e[2]]
# a proof has ONE k_grind launch (the final PoW; the interaction PoW is part of k_step_pow_relations).  bench.py's LAST proof is
# the verification proof made on the main host thread (cold device pool: hipMallocs), so the proof analysed is the one before
# it: the last TIMED proof.
def proof_start(after_grind):
    i = after_grind + 1
    while i < len(ev) and ('gather' in ev[i][2] or 'COPY' in ev[i][2] or 'copyBuffer' in ev[i][2]) and 'k_preproc' not in ev[i][2]:
        i += 1
    return i
import os
g = int(os.environ.get("GRINDS_PER_PROOF", "1"))   # 2 for library builds older than the device-side interaction PoW
i0, i1 = proof_start(gi[-(2 * g + 1)]), proof_start(gi[-(g + 1)])
sub = ev[i0:i1]
t0 = sub[0][0]
def cover(events):
    """union of the intervals: (end of the last one, busy time, gaps)"""
    cur, busy, gaps, prev = t0, 0, [], None
    for s, e, n, q, g in events:
        if s > cur:
            gaps.append((s - cur, (cur - t0) / 1e3, prev, n))
            busy += e - s
            cur = e
        elif e > cur:
            busy += e - cur
            cur = e
        prev = n
    return cur, busy, gaps
# Three readings of "busy": every dispatch and copy; without the fork / join spin-wait kernels (a k_join_collect lane POLLS a flag
# word until another stream's work is done: the GPU is waiting, not working); and without the staging copies as well
SPIN = ('k_join_flag', 'k_join_collect')
STAGE = ('k_stage_copy', 'COPY', 'copyBuffer', 'fillBuffer')
end_all, busy_all, gaps = cover(sub)
work = [x for x in sub if x[2] not in SPIN]
_, busy_work, gaps_work = cover(work)
_, busy_kern, _ = cover([x for x in work if not any(t in x[2] for t in STAGE)])
span = end_all - t0
print(f"span {span / 1e6:.3f} ms  busy {busy_all / 1e6:.3f} ms  idle {(span - busy_all) / 1e6:.3f} ms  events {len(sub)}")
print(f"without the fork/join spin-wait kernels ({sum(1 for x in sub if x[2] in SPIN)} launches, {sum(x[1] - x[0] for x in sub if x[2] in SPIN) / 1e6:.3f} ms summed): "
      f"busy {busy_work / 1e6:.3f} ms  idle {(span - busy_work) / 1e6:.3f} ms;  without the staging copies too: busy {busy_kern / 1e6:.3f} ms  idle {(span - busy_kern) / 1e6:.3f} ms")
gaps = gaps_work
for g in sorted(gaps, reverse=True)[:25]:
    print(f"gap {g[0] / 1e3:8.1f} us at {g[1]:9.1f} us  after {g[2]}  before {g[3]}")
if '--list' in sys.argv:
    p = t0
    for s, e, n, q, g in sub:
        print(f"{(s - t0) / 1e3:9.1f} us gap {(s - p) / 1e3:7.1f} dur {(e - s) / 1e3:7.1f} q{q} {n} {g}")
        p = max(p, e)
