"""bench.py's output contract on a small workload (the driver runs the default workload itself): one JSON line with
BASELINE.json's metric / unit, the whole-job `value`, `roofline` (dominant kernel, HIP-event timed, HBM bound, traffic from
the committed PMC profile), `cpu_baseline` (the oracle on the host cores, kind "port") and the parity flag; `--gpus 2`
without a launcher spawns its own ranks (here: two ranks sharing the GPU over gloo) and reports the sharded mode too."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    return json.loads(lines[0])


def test_single_gpu_line():
    d = _bench("--steps", "2", "--warmup", "1", "--fib-n", "3000", "--cpu-sample-n", "3000", "--pipelined", "2", "--big-fib-n", "6000")
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base["metric"] and d["unit"] == "M31 trace cells/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "u32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["cells_per_proof"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    # the dominant class is Blake2s or butterflies: integer-VALU-bound ("valu", with the ALU fraction in `alu`); the HBM figures
    # (algorithmic bytes / HIP-event time against 8 TB/s) stay beside it
    assert r["bound"] in ("valu", "hbm") and (r["bound"] == "hbm" or r["alu"]["frac"] > 0)
    assert r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert "gpu_idle_ms" not in d and (d["gpu_idle_traced"] is None or d["gpu_idle_traced"]["idle_ms"] >= 0)
    big = r["whole_path_model_larger_sizes"]
    assert big and "error" not in big[0] and big[0]["whole_path_model"]["frac"] > 0
    assert "traffic" in r and r["launches"] > 0 and r["avg_launch_ms"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == d["unit"] and "sample" in c
    assert d["parity_at_metric_config"] is True and d["proof_verified"] is True       # sample == bench workload here
    assert d["end_to_end"]["host_prover_input_ms"]["min"] > d["ms_per_step"] * 0.5
    # round 4: per-stage roofline against SURVEY 8d's bytes-per-cell table, the measurement set the traffic figure comes from, and
    # the streaming-ingest legs next to the resident pipeline
    st = {x["stage"]: x for x in r["stages"]}
    assert {"IFFT + LDE", "Merkle hashing", "constraint quotients", "DEEP quotients"} <= set(st)
    assert abs(sum(x["bytes_per_cell"] for x in r["stages"]) - 52.0) < 1e-9 and all(0 < x["frac"] < 1 for x in r["stages"])
    latest = open(os.path.join(ROOT, "profiles", "LATEST")).read().split()[0]
    assert r["traffic_source"] is None or os.path.basename(r["traffic_source"]).startswith(latest + "_")
    # round 6: the same segment through cm_prove_sharded with one rank (in-library RCCL communicator), next to the single-GPU prover
    so = d["sharded_one_rank"]
    assert "error" not in so, so
    assert so["bit_identical_to_single_gpu_proof"] is True and so["world"] == 1 and so["ms_per_proof"] > 0 and so["single_gpu_ms_same_process"] > 0
    e = d["end_to_end"]
    assert e["pipelined_from_host_ms_per_proof"] > 0 and e["pipelined_from_segments_ms_per_proof"] > 0 and e["streamed_vs_resident"] > 0.5


def test_gpus_2_self_spawns_and_reports_both_modes():
    d = _bench("--gpus", "2", "--dist-backend", "gloo", "--force-device", "0", "--steps", "2", "--warmup", "1", "--fib-n", "3000",
               "--no-cpu-baseline", "--pipelined", "0")
    assert d["n_gpus"] == 2 and d["cpu_baseline"] is None
    assert abs(d["value"] - 2 * d["config"]["cells_per_proof"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]     # whole-job aggregate
    s = d["sharded"]
    assert "error" not in s, s
    assert s["bit_identical_to_single_gpu_proof"] is True and s["ms_per_proof"] > 0 and len(s["component_owner"]) == 34


def test_a_stuck_sharded_child_costs_only_the_sharded_object():
    # a 1-second limit is shorter than the child's start-up: the child job is killed, the replica line must still be complete
    d = _bench("--gpus", "2", "--dist-backend", "gloo", "--force-device", "0", "--steps", "1", "--warmup", "1", "--fib-n", "3000",
               "--no-cpu-baseline", "--pipelined", "0", "--sharded-timeout", "1")
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["proof_verified"] is True
    assert "error" in d["sharded"] and "killed" in d["sharded"]["error"]
