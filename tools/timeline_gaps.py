#!/usr/bin/env python3
"""GPU-idle gaps of the LAST proof in a rocprofv3 kernel + memory-copy trace (csv output):
   rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -- python bench.py --steps 3 --warmup 2 --pipelined 0 --no-kprof
   tools/timeline_gaps.py DIR/<host>/<pid>_kernel_trace.csv DIR/<host>/<pid>_memory_copy_trace.csv [--list]
Prints span / busy / idle and the largest gaps with the events around them (development tool)."""
import csv, re, sys
k, m = sys.argv[1], sys.argv[2]
ev = []
for r in csv.DictReader(open(k)):
    n = re.sub(r'\(.*', '', r['Kernel_Name']).replace('cm::', '').replace('void ', '')[:44]
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
cur, busy, gaps, prev = t0, 0, [], None
for s, e, n, q, g in sub:
    if s > cur:
        gaps.append((s - cur, (cur - t0) / 1e3, prev, n))
        busy += e - s
        cur = e
    elif e > cur:
        busy += e - cur
        cur = e
    prev = n
print(f"span {(cur - t0) / 1e6:.3f} ms  busy {busy / 1e6:.3f} ms  idle {(cur - t0 - busy) / 1e6:.3f} ms  events {len(sub)}")
for g in sorted(gaps, reverse=True)[:25]:
    print(f"gap {g[0] / 1e3:8.1f} us at {g[1]:9.1f} us  after {g[2]}  before {g[3]}")
if '--list' in sys.argv:
    p = t0
    for s, e, n, q, g in sub:
        print(f"{(s - t0) / 1e3:9.1f} us gap {(s - p) / 1e3:7.1f} dur {(e - s) / 1e3:7.1f} q{q} {n} {g}")
        p = max(p, e)
