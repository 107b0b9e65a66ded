# development tool: two or more BUILDS of the library timed alternately in ONE GPU session (boxes differ by +-4 %): lone proofs and
# proofs with four in flight.   python tools/ab_libs.py [--rounds 3] libA.so libB.so ...
# Every timing is a fresh process (CAIROM_HIP_LIB selects the build); prints lone ms per proof, the phase times and the pipelined rate.
import argparse, json, os, statistics, subprocess, sys
ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--pipelined", type=int, default=4)
a = ap.parse_args()
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
res = {l: {"lone": [], "pipe": [], "phases": []} for l in a.libs}
for r in range(a.rounds):
    for l in (a.libs if r % 2 == 0 else a.libs[::-1]):
        env = dict(os.environ, CAIROM_HIP_LIB=os.path.abspath(l))
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "16", "--warmup", "4", "--no-cpu-baseline", "--no-end-to-end",
                              "--alt-fib-n", "0", "--big-fib-n", "0", "--cached-setup-steps", "0", "--sharded-one-rank-blocks", "0", "--no-kprof", "--pipelined", str(a.pipelined)], env=env, capture_output=True, text=True, cwd=root)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception:
            print(l, "FAILED", out.stderr[-400:]); continue
        res[l]["lone"].append(d["ms_per_step"])
        if d.get("pipelined"): res[l]["pipe"].append(d["pipelined"]["ms_per_proof"])
        res[l]["phases"].append(d["phase_ms"])
        print(f"{os.path.basename(l):28s} lone {d['ms_per_step']:.3f} ms  pipelined {d['pipelined']['ms_per_proof'] if d.get('pipelined') else float('nan'):.3f} ms", flush=True)
for l, v in res.items():
    if not v["lone"]: continue
    ph = {k: round(statistics.mean(p[k] for p in v["phases"]), 3) for k in v["phases"][0]}
    print(f"{os.path.basename(l):28s} lone median {statistics.median(v['lone']):.3f}  pipelined median {statistics.median(v['pipe']) if v['pipe'] else float('nan'):.3f}  phases {ph}")
