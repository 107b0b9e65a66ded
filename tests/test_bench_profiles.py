"""CPU test, no GPU: the committed measurement set that `bench.py` cites (profiles/LATEST) is complete and the figures bench.py derives
from it are recomputable — `roofline.valu_model` from the PMC instruction counts x the static instruction-mix table, `gpu_idle_traced`
from the timeline listing, `roofline.traffic` from the PMC traffic file."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_latest_profile_set_is_complete():
    b = _bench()
    tag = open(os.path.join(ROOT, "profiles", "LATEST")).read().split()[0]
    for suffix in ("bench_n1.json", "rocprofv3_kernel_stats.csv", "pmc_traffic.json", "pmc_sq.json", "valu_mix.json", "gaps.txt"):
        assert b.latest_profile(suffix), f"profiles/{tag}_{suffix} is missing"


def test_valu_model_is_recomputable_from_the_committed_files():
    b = _bench()
    v = b.valu_model(9.2)
    assert v and v["floor_ms"] > 0 and 0.2 < v["frac"] < 1.0
    sq = json.load(open(os.path.join(ROOT, v["sources"][0])))["classes"]
    mix = json.load(open(os.path.join(ROOT, v["sources"][1])))
    # recompute by hand: sum over classes of SQ_INSTS_VALU x mean ns / 1024 SIMDs
    total = 0.0
    for k, c in sq.items():
        n = c.get("SQ_INSTS_VALU", 0.0)
        if not n or not k:
            continue
        ns = mix["classes"].get(k, {}).get("mean_ns_per_wave_instruction", mix["default_mean_ns_per_wave_instruction"])
        total += n * ns / 1024.0
    assert abs(total * 1e-6 - v["floor_ms"]) < 1e-9
    # the buckets' costs are the lab's lane-op rates (64 lanes x 1024 SIMDs / rate)
    assert abs(mix["ns_per_wave_instruction"]["full"] - 64 * 1024 / 65.9e12 * 1e9) < 1e-9
    assert {s["class"] for s in v["stages"]} >= {"k_merkle_layer", "k_fft_fused_rb", "k_constraints(region)", "k_logup(region)", "k_quotients"}


def test_traced_idle_reports_the_spin_free_figures():
    b = _bench()
    t = b.traced_idle()
    assert t and t["idle_ms"] <= t["idle_ms_without_spin_wait_kernels"] <= t["idle_ms_without_spin_wait_and_staging"] < 2.0
    assert t["spin_wait_launches"] > 0


def test_dominant_class_traffic_is_on_file():
    b = _bench()
    traffic, src = b.pmc_traffic("k_merkle_layer")
    assert traffic and traffic > 1e6 and src.startswith("profiles/")
