#!/usr/bin/env python3
"""HBM traffic per kernel class from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass:
TCC has 4 slots, FETCH_SIZE takes 3 and WRITE_SIZE 2 — /opt/skills/guides/MI355X_MICROARCH.md §rocprofv3 PMC slots).

usage: pmc_traffic.py <fetch_results.db> <write_results.db> --proofs N [--json out.json]

  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o f -- python bench.py --steps 1 --warmup 1 ...
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write -o w -- python bench.py --steps 1 --warmup 1 ...

Both counters are in KiB.  gfx950 correction (same guide, §HBM): FETCH_SIZE reports half of the bytes of a
coalesced streaming read, so  hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  The factor is calibrated in
the guide for 16-B/lane loads; for this library's dword-per-lane column reads the check is the write side
(WRITE_SIZE of the FFT passes equals the algorithmic 4 B/element) and the Merkle leaf reads (2 * FETCH_SIZE
= the column bytes within a few per cent).

Kernel classes are the names bench.py's in-library timing uses (kprof): all instances of a template are
one class.
"""
import argparse
import collections
import json
import re
import sqlite3
import sys

CLASSES = [
    (r"k_fft_pass(_r8|_rb)?<(false|0)", "k_fft_pass<fft>"),
    (r"k_fft_pass(_r8|_rb)?<(true|1)", "k_fft_pass<ifft>"),
    (r"k_merkle_layer|k_merkle_narrow", "k_merkle_layer"),   # one kprof class: the narrow-layer kernel serves the same launches
    (r"k_merkle_multi", "k_merkle_multi"),
    (r"k_merkle_tail", "k_merkle_tail"),
    (r"k_quotients", "k_quotients"),
    (r"k_constraints", "k_constraints(region)"),
    (r"k_logup", "k_logup(region)"),
    (r"k_opcode_trace|k_hist|k_memory_trace|k_merkle_trace|k_clock_update_trace|k_poseidon2_trace", "k_trace_gen(region)"),
    (r"k_eval_partial_multi|k_point_tables_multi|k_reduce_partials_multi|k_eval_at_point", "k_eval_at_point"),
    (r"k_fri_tail", "k_fri_tail"),
]


def klass(name):
    n = name.replace("void ", "").replace("cm::", "")
    for pat, c in CLASSES:
        if re.match(pat, n):
            return c
    return n.split("(")[0].split("<")[0]


def per_class(db, counter):
    c = sqlite3.connect(db)
    # one row per (dispatch, counter[, dimension]); sum the dimensions of a dispatch
    q = ("select k.name, p.dispatch_id, sum(p.counter_value) from pmc_events p join kernels k on p.dispatch_id = k.dispatch_id "
         "where p.counter_name = ? group by p.dispatch_id")
    agg = collections.defaultdict(lambda: [0, 0.0])
    for name, _, v in c.execute(q, (counter,)):
        a = agg[klass(name)]
        a[0] += 1
        a[1] += v
    return agg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_db")
    ap.add_argument("write_db")
    ap.add_argument("--proofs", type=int, required=True, help="proofs in each profiled run (warmup + instrumented + steps)")
    ap.add_argument("--json")
    a = ap.parse_args()
    f = per_class(a.fetch_db, "FETCH_SIZE")
    w = per_class(a.write_db, "WRITE_SIZE")
    out = {}
    for k in sorted(set(f) | set(w), key=lambda k: -(2 * f[k][1] + w[k][1])):
        launches = max(f[k][0], w[k][0])
        if launches == 0:
            continue
        fetch_kib, write_kib = f[k][1], w[k][1]
        hbm = (2.0 * fetch_kib + write_kib) * 1024.0
        out[k] = {"launches_per_proof": launches / a.proofs, "fetch_KiB_raw_per_proof": fetch_kib / a.proofs,
                  "write_KiB_per_proof": write_kib / a.proofs, "hbm_bytes_per_proof": hbm / a.proofs,
                  "hbm_bytes_per_launch": hbm / launches}
    for k, v in list(out.items())[:16]:
        print(f"{k:28s} launches/proof={v['launches_per_proof']:7.1f} fetch(raw)={v['fetch_KiB_raw_per_proof'] / 1e6:7.3f} GiB*  "
              f"write={v['write_KiB_per_proof'] / 1e6:7.3f}  hbm={v['hbm_bytes_per_proof'] / 1e9:7.3f} GB/proof")
    if a.json:
        json.dump({"correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950, MI355X_MICROARCH.md §HBM)",
                   "proofs_in_run": a.proofs, "classes": out}, open(a.json, "w"), indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
