#!/usr/bin/env python3
"""Per-kernel-class SQ counters from one rocprofv3 PMC pass (issue-bound or stall-bound?).

usage: pmc_sq.py <results.db> --proofs N [--json out.json]

  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU \
            SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY -d gpurun_out/pmc_sq -o q -- python bench.py --steps 1 ...

Units (/opt/skills/guides/MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over
waves; WAIT_ANY (wave parked on s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY ~ WAVE_CYCLES.
Derived per class:
  valu_active_frac = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES   share of wave time with a VALU instruction executing
  wait_frac        = SQ_WAIT_ANY / SQ_WAVE_CYCLES           share parked on memory / barriers
  issue_stall_frac = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
  valu_per_wave_cycle = SQ_INSTS_VALU / SQ_WAVE_CYCLES
A class is issue-bound when valu_active_frac (x waves per SIMD) saturates the SIMD and wait_frac is small."""
import argparse
import collections
import json
import sqlite3
import sys

from pmc_traffic import klass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--proofs", type=int, required=True)
    ap.add_argument("--json")
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    q = ("select k.name, p.counter_name, sum(p.counter_value), count(distinct p.dispatch_id) from pmc_events p "
         "join kernels k on p.dispatch_id = k.dispatch_id group by k.name, p.counter_name")
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(int)
    for name, cn, v, nd in c.execute(q):
        kc = klass(name)
        agg[kc][cn] += v
        if cn == "SQ_WAVE_CYCLES":
            launches[kc] += nd
    out = {}
    for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        wc = d.get("SQ_WAVE_CYCLES", 0) or 1.0
        out[k] = {"launches_per_proof": launches[k] / a.proofs, **{cn: v / a.proofs for cn, v in d.items()},
                  "valu_active_frac": d.get("SQ_ACTIVE_INST_VALU", 0) / wc, "wait_frac": d.get("SQ_WAIT_ANY", 0) / wc,
                  "issue_stall_frac": d.get("SQ_WAIT_INST_ANY", 0) / wc, "any_active_frac": d.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                  "valu_insts_per_wave_quadcycle": d.get("SQ_INSTS_VALU", 0) / wc,
                  "salu_per_valu": d.get("SQ_INSTS_SALU", 0) / (d.get("SQ_INSTS_VALU", 0) or 1.0)}
    print(f"{'class':28s} {'wave_cyc/proof':>14s} {'valu_act':>8s} {'any_act':>8s} {'wait':>6s} {'stall':>6s} {'valu/qc':>8s} {'salu/valu':>9s}")
    for k, v in list(out.items())[:18]:
        print(f"{k:28s} {v.get('SQ_WAVE_CYCLES', 0):14.3e} {v['valu_active_frac']:8.3f} {v['any_active_frac']:8.3f} {v['wait_frac']:6.3f} "
              f"{v['issue_stall_frac']:6.3f} {v['valu_insts_per_wave_quadcycle']:8.3f} {v['salu_per_valu']:9.3f}")
    if a.json:
        json.dump({"proofs_in_run": a.proofs, "units": "quad-cycles summed over waves (SQ_*), per proof", "classes": out}, open(a.json, "w"), indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
