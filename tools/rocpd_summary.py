#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (`*_results.db`) into the CSV kept under profiles/.

usage: rocpd_summary.py <results.db> [--csv out.csv] [--timeline] [--pmc]

* default: per-kernel calls / total / average duration (the `--stats` table), plus GPU busy time vs
  wall time of the kernel span (launch gaps + host round trips = wall - busy).
* --timeline: every dispatch of the LAST proof in the trace (start offset, kernel, grid, duration).
* --pmc: per-kernel sums of the collected counters (one row per kernel x counter).
"""
import argparse
import collections
import sqlite3
import sys


def short(name):
    n = name.replace("(anonymous namespace)::", "").split("(")[0]
    return n.replace("void ", "").replace("cm::", "")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--csv")
    ap.add_argument("--timeline", action="store_true")
    ap.add_argument("--pmc", action="store_true")
    ap.add_argument("--header", default="")
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    rows = c.execute("select name,start,end,grid_x,workgroup_x,vgpr_count,lds_size from kernels order by start").fetchall()
    if not rows:
        print("no kernel dispatches in", a.db)
        return 1
    agg = collections.defaultdict(lambda: [0, 0])
    for n, s, e, *_ in rows:
        k = agg[short(n)]
        k[0] += 1
        k[1] += e - s
    busy = sum(e - s for _, s, e, *_ in rows)
    wall = rows[-1][2] - rows[0][1]
    out = []
    if a.header:
        out.append("# " + a.header)
    out.append(f"# dispatches={len(rows)} gpu_busy_ms={busy / 1e6:.3f} span_ms={wall / 1e6:.3f} (span - busy = launch gaps + host round trips)")
    out.append("name,calls,total_us,avg_us,percent")
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.append(f'"{k}",{n},{t / 1e3:.3f},{t / 1e3 / n:.3f},{100.0 * t / busy:.3f}')
    text = "\n".join(out) + "\n"
    if a.csv:
        open(a.csv, "w").write(text)
    print(text)
    if a.timeline:
        # last proof = dispatches after the last k_preproc burst start (first kernel of a proof)
        starts = [i for i, r in enumerate(rows) if "k_preproc" in r[0] and (i == 0 or "k_preproc" not in rows[i - 1][0])]
        i0 = starts[-1] if starts else 0
        t0 = rows[i0][1]
        prev_end = t0
        for n, s, e, g, wg, vg, lds in rows[i0:]:
            print(f"{(s - t0) / 1e6:9.3f} gap={(s - prev_end) / 1e3:7.1f}us {short(n)[:48]:48s} grid={g:9d} wg={wg:4d} vgpr={vg:3d} lds={lds:6d} {(e - s) / 1e3:8.1f} us")
            prev_end = e
    if a.pmc:
        try:
            q = ("select k.name, p.counter_name, sum(p.counter_value), count(*) from pmc_events p join kernels k on p.dispatch_id = k.dispatch_id "
                 "group by k.name, p.counter_name")
            for n, cn, v, cnt in c.execute(q):
                print(f"{short(n)[:50]:50s} {cn:24s} sum={v:.6g} n={cnt}")
        except sqlite3.Error as ex:
            print("pmc query failed:", ex)
            cur = c.execute("select * from pmc_events limit 1")
            print([d[0] for d in cur.description])
    return 0


if __name__ == "__main__":
    sys.exit(main())
