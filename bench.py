#!/usr/bin/env python3
"""bench.py — throughput of the hot path (prove one Cairo-M segment) on MI355X.

Metric (BASELINE.json): M31 trace cells/s proved, fibonacci_loop at ~2^22 rows, end-to-end proof ms.
A "step" = one whole proof (trace gen -> proof object) of the synthetic fibonacci_loop segment with the
ProverInput already resident in HBM.  N > 1: one process per GPU (torch.distributed / RCCL used only for
the barrier and the max-over-ranks reduction); every rank proves an independent segment (continuation
segments are independent proofs — SURVEY §8e-1), so scaling is weak and there is no data-path collective.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP-event timed inside the library on
its launch stream) and `cpu_baseline` (the CPU oracle on a bounded sample, rank 0 / N=1 only).
"""
import argparse
import ctypes as C
import datetime
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
ALL_CPUS = os.sched_getaffinity(0)   # what the oracle's child process gets, whatever this process does later

FIB_N = 419_000          # 10*n + 12 = 4,190,012 VM steps (~2^22), one segment
HBM_PEAK_GBS = 8000.0    # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# Integer-ALU ceilings of the two dominant kernel classes, from the chip-wide lane-op rates tools/valu_lab.hip measures on gfx950
# (profiles/r03k_valu_lab.txt, no clock assumption): VOP2 v_xor 65.9 T, v_add 64.0 T lane-ops/s; VOP3 v_alignbit 37.9 T,
# v_add3 35.85 T; v_mad_u64_u32 31.46 T.  A Blake2s compression is 336 xor + 160 add + 320 alignbit + 160 add3 per lane
# (the plain op mix; the shipped kernel replaces 80 xor + 80 alignbit by 160 SDWA xors, which the lab rates would price
# HIGHER although the kernel measures 5 % faster — the ceiling keeps the plain mix); an M31 butterfly is one mad_u64 + ~10 VOP2.
_VOP2, _ADD, _ALIGN, _ADD3, _MAD64 = 65.9e12, 64.0e12, 37.9e12, 35.85e12, 31.46e12
_B2S_PEAK = 1.0 / (336 / _VOP2 + 160 / _ADD + 320 / _ALIGN + 160 / _ADD3)       # ~4.9e10 compressions/s
_BFLY_PEAK = 1.0 / (1 / _MAD64 + 10 / _VOP2)
# What the hardware actually sustains on that op mix: the per-lane compression looping on registers only (no loads, no stores) on
# every CU at 8 waves per SIMD — tools/chain_lab.hip, profiles/r03w_chain_lab.txt: 3.78e10 compressions/s in the order the compiler
# emits, 4.08e10 with the four G functions of a half-round issued in lockstep (the best order found; in the Merkle kernels
# themselves that order changes nothing).  The single-op lane rates above do not add up when VOP2 / VOP3 / SDWA ops alternate and
# when dependent instructions follow each other (profiles/r03w_valu_lab.txt, the "mix" lines).
B2S_REGISTER_ONLY = 4.08e10
ALU_PEAK = {"k_merkle_layer": (_B2S_PEAK, "Blake2s compressions/s"),
            "k_fft_pass<fft>": (_BFLY_PEAK, "M31 butterflies/s"),
            "k_fft_pass<ifft>": (_BFLY_PEAK, "M31 butterflies/s")}
MODEL_BYTES_PER_CELL = 52.0  # SURVEY §8d algorithmic-bytes model for the whole path


def pmc_traffic(kernel_class):
    """HBM bytes per launch of one kernel class from the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE need separate passes, so bench.py cannot collect them live; tools/pmc_traffic.py makes the
    file from `rocprofv3 --pmc` runs of this same command).  None when no file covers the class."""
    f = latest_profile("pmc_traffic.json")
    if not f:
        return None, None
    try:
        d = json.load(open(f))
        c = d["classes"].get(kernel_class)
        return (c["hbm_bytes_per_launch"], os.path.relpath(f, ROOT)) if c else (None, None)
    except (OSError, ValueError, KeyError):
        return None, None


def traced_idle():
    """GPU-idle time inside one proof from the committed kernel + copy timeline of the measurement set profiles/LATEST names
    (tools/gaps.sh: rocprofv3 --kernel-trace --memory-copy-trace of this bench, tools/timeline_gaps.py) — the sum of the intervals
    in which no kernel and no copy of the proof runs.  Not measured in this run (it needs the tracer)."""
    f = latest_profile("gaps.txt")
    if not f:
        return None
    try:
        import re
        lines = open(f).read().splitlines()
        m = re.match(r"span ([\d.]+) ms\s+busy ([\d.]+) ms\s+idle ([\d.]+) ms", lines[0])
        out = {"span_ms": float(m.group(1)), "busy_ms": float(m.group(2)), "idle_ms": float(m.group(3)), "source": os.path.relpath(f, ROOT)}
        # (round 6) the same timeline with the fork / join spin-wait kernels — and the staging copies — not counted as work
        m2 = re.match(r"without the fork/join spin-wait kernels \((\d+) launches, ([\d.]+) ms summed\): busy ([\d.]+) ms\s+idle ([\d.]+) ms;"
                      r"\s+without the staging copies too: busy ([\d.]+) ms\s+idle ([\d.]+) ms", lines[1]) if len(lines) > 1 else None
        if m2:
            out.update({"spin_wait_launches": int(m2.group(1)), "spin_wait_kernel_ms": float(m2.group(2)),
                        "idle_ms_without_spin_wait_kernels": float(m2.group(4)), "idle_ms_without_spin_wait_and_staging": float(m2.group(6)),
                        "note": "idle_ms counts every dispatch as busy, including the one-wave kernels that only poll a fork / join flag; "
                                "the two other figures leave those (and the staging copies) out"})
        return out
    except (OSError, AttributeError, ValueError):
        return None


def latest_profile(suffix):
    """profiles/<tag>_<suffix> of the measurement set of HEAD: the tag is the first word of profiles/LATEST (written when a set is
    committed; file names do not sort by age — r03zz sorts behind r03h).  None when that set has no such file."""
    try:
        tag = open(os.path.join(ROOT, "profiles", "LATEST")).read().split()[0]
    except (OSError, IndexError):
        return None
    f = os.path.join(ROOT, "profiles", f"{tag}_{suffix}")
    return f if os.path.exists(f) else None


def valu_model(ms_per_step):
    """Whole-path VALU roofline (round 6): the time the SIMDs' vector ALU ports need AT LEAST for the instructions one proof issues.
    Per kernel class: SQ_INSTS_VALU (wave-instructions per proof, PMC pass of the measurement set profiles/LATEST names) x the mean
    issue cost of that class's instruction mix (profiles/<tag>_valu_mix.json: static ISA of the class's kernels in three buckets priced
    by the single-opcode lab profiles/r03k_valu_lab.txt) / 1024 SIMDs.  floor_ms = sum over classes; frac = floor_ms / ms_per_step.
    Everything is recomputable from the two committed files; nothing here is measured in this run except ms_per_step."""
    fsq, fmix = latest_profile("pmc_sq.json"), latest_profile("valu_mix.json")
    if not fsq or not fmix:
        return None
    try:
        sq, mix = json.load(open(fsq))["classes"], json.load(open(fmix))
    except (OSError, ValueError, KeyError):
        return None
    stages, total_ns, total_insts = [], 0.0, 0.0
    for k, c in sorted(sq.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0.0)):
        n = c.get("SQ_INSTS_VALU", 0.0)
        if not n or not k:
            continue
        m = mix["classes"].get(k)
        ns = (m or {}).get("mean_ns_per_wave_instruction", mix["default_mean_ns_per_wave_instruction"])
        t = n * ns / 1024.0
        total_ns += t
        total_insts += n
        if n >= 2e6:
            stages.append({"class": k, "valu_wave_insts": n, "mean_ns_per_wave_inst": ns, "mix": "own" if m else "default", "floor_ms": t * 1e-6,
                           "valu_active_x_waves": c.get("valu_insts_per_wave_quadcycle"), "wait_frac": c.get("wait_frac")})
    return {"unit": "ms per proof", "floor_ms": total_ns * 1e-6, "frac": total_ns * 1e-6 / ms_per_step, "valu_wave_insts_per_proof": total_insts,
            "simds": 1024, "stages": stages, "sources": [os.path.relpath(fsq, ROOT), os.path.relpath(fmix, ROOT), "profiles/r03k_valu_lab.txt"],
            "note": "lower bound of the proof time if every SIMD issued VALU instructions back to back at the lab's single-opcode rates and "
                    "nothing else cost anything; `frac` is the share of the measured proof time that bound explains (the HBM fractions stay "
                    "in `frac`, `stages`, `whole_path_model`).  The Blake2s and butterfly kernels cannot reach the single-opcode rates in a "
                    "mixed stream (register-only compression loop: 0.84 of them, profiles/r03w_chain_lab.txt), so 1.0 is not attainable."}


# algorithmic bytes per committed trace cell of every stage (each stage streams its input once and writes its output
# once) and the kprof classes whose HIP-event intervals make up the stage.  The class sets are wider than the model's stages:
# the transforms include the constraint accumulators' and the composition polynomial's, the Merkle classes every FRI tree and
# the composition tree (the model prices those at ~0), so the fractions are lower bounds of what the model's stage reaches.
STAGE_MODEL = [
    ("trace / interaction generation", 4.0, ("k_trace_gen(region)", "k_logup(region)")),
    ("IFFT + LDE", 8.0 + 12.0, ("k_fft_pass<ifft>", "k_fft_pass<fft>", "k_fft_fused", "k_small_commit")),
    ("Merkle hashing", 8.0, ("k_merkle_layer", "k_merkle_layer_quad", "k_merkle_multi", "k_merkle_top", "k_merkle_tail", "k_fold_leaf")),
    ("constraint quotients", 8.0, ("k_constraints(region)",)),
    ("OODS eval_at_point", 4.0, ("k_eval_at_point",)),
    ("DEEP quotients", 8.0, ("k_quotients",)),
]


def stage_roofline(kprof_all, n_prof, cells):
    out = []
    for name, bpc, classes in STAGE_MODEL:
        ms = sum(kprof_all[c]["ms"] for c in classes if c in kprof_all) / n_prof
        if ms <= 0:
            continue
        gbs = bpc * cells / (ms * 1e-3) / 1e9
        out.append({"stage": name, "bytes_per_cell": bpc, "ms_per_step": ms, "achieved_GBs": gbs, "frac": gbs / HBM_PEAK_GBS,
                    "classes": [c for c in classes if c in kprof_all]})
    return out


def _oracle_child(fib_n, threads, reps, words_out=None, timeout=900):
    """One run of oracle/cpu_baseline.py in a child process with an explicit OpenMP placement: the host's full CPU set,
    `threads` threads bound to consecutive physical cores (OMP_PROC_BIND=close, OMP_PLACES=cores) so that the team, and the
    memory its master first-touches, sit on one socket whatever the parent process did before."""
    import subprocess
    env = dict(os.environ)
    env.update({"OMP_NUM_THREADS": str(threads), "OMP_PROC_BIND": "close", "OMP_PLACES": "cores", "OMP_DYNAMIC": "false",
                "CM_CPU_AFFINITY": "0", "PYTHONPATH": ROOT + os.pathsep + env.get("PYTHONPATH", "")})
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), "--fib-n", str(fib_n), "--reps", str(reps)]
    if words_out:
        cmd += ["--words-out", words_out]
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout,
                       preexec_fn=lambda: os.sched_setaffinity(0, ALL_CPUS))
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        raise RuntimeError(f"oracle/cpu_baseline.py failed ({r.returncode}): {r.stderr[-800:]}")
    return json.loads(lines[-1])


def cpu_baseline(sample_n, single_n, threads, hip_words=None):
    """Oracle (CPU restatement, 'port') on bounded samples of the same workload family, SURVEY §8d: all-thread run (median of
    two) on `sample_n` and a single-thread run on `single_n`.  When `sample_n` IS the bench workload (the default) the
    oracle's proof words are compared with the HIP proof of the same ProverInput: the returned `parity` is True / False, or
    None when the sample differs from the bench workload."""
    import subprocess
    import tempfile
    import numpy as np
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    with tempfile.TemporaryDirectory() as td:
        wpath = os.path.join(td, "oracle_words.npy")
        multi = _oracle_child(sample_n, threads, 2, wpath)
        words = np.load(wpath)
    single = _oracle_child(single_n, 1, 1) if single_n > 0 else None
    parity = None
    if hip_words is not None:
        parity = bool(words.size == hip_words.size and np.array_equal(words, hip_words))
    dt = sorted(multi["seconds"])[len(multi["seconds"]) // 2] if len(multi["seconds"]) % 2 else sum(sorted(multi["seconds"])[:2]) / 2
    cells, steps = multi["cells"], multi["steps"]
    out = {"value": cells / dt, "unit": "M31 trace cells/s", "cores": threads, "kind": "port",
           "sample": f"fibonacci_loop n={sample_n} ({steps} VM steps, {cells} cells incl. the fixed 2^20/2^18/2^16 preprocessed + "
                     f"range-check tables), oracle prove_segment (own CPU restatement, NOT Stwo SimdBackend), OpenMP x{threads} "
                     f"(OMP_PROC_BIND=close, OMP_PLACES=cores) of {os.cpu_count()} host CPUs, median of "
                     f"{', '.join(f'{t:.1f}' for t in multi['seconds'])} s",
           "seconds": multi["seconds"],
           "reference_slot": {"value": None, "unit": "M31 trace cells/s",
                              "how": "RUSTFLAGS='-C target-cpu=native' cargo bench --bench prover_speed_benchmark "
                                     "(crates/prover/benches/prover_speed_benchmark.rs:37-72) wherever Rust nightly-2025-04-06 + the "
                                     "stwo submodule exist; neither does in this image"}}
    if single:
        s_dt = single["seconds"][0]
        out["single_thread"] = {"value": single["cells"] / s_dt, "unit": "M31 trace cells/s", "cores": 1,
                                "sample": f"fibonacci_loop n={single_n} ({single['steps']} VM steps, {single['cells']} cells; "
                                          f"BASELINE configs[0] when n = 1000), {s_dt:.1f} s"}
    return parity, out


def run_sharded_child(args, world, comm="torch", timeout=None):
    """`python -m torch.distributed.run ... -m cairo_m_amd.sharded --json` with a time limit; returns bench.py's `sharded` object.
    comm: "torch" = collectives through torch.distributed callbacks (host-blocking); "rccl" = the library's own stream-ordered
    RCCL communicator (cm_rccl_comm_create)."""
    timeout = timeout or args.sharded_timeout
    import signal
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items()
           if not (k.startswith("TORCHELASTIC_") or k.startswith("ROLE_") or k.startswith("GROUP_") or
                   k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"))}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "cairo_m_amd.sharded", "--fib-n", str(args.fib_n), "--steps", str(args.steps),
           "--dist-backend", args.dist_backend, "--check-single", "--json", "--comm", comm]
    if args.force_device >= 0:
        cmd += ["--force-device", str(args.force_device)]
    mode = "one proof sharded over all ranks (strong scaling)"
    import tempfile
    with tempfile.TemporaryFile("w+") as fo, tempfile.TemporaryFile("w+") as fe:     # files, not pipes: nothing can block on them
        p = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=fo, stderr=fe, start_new_session=True,
                             preexec_fn=lambda: os.sched_setaffinity(0, ALL_CPUS))   # every rank places itself next to ITS GPU
        try:
            p.wait(timeout=timeout)
        except subprocess.TimeoutExpired:
            # the launcher and every rank it started (the ranks sit in process groups of their own): exactly those PIDs
            import psutil
            procs = []
            try:
                procs = psutil.Process(p.pid).children(recursive=True)
            except psutil.Error:
                pass
            for q in procs:
                try:
                    q.kill()
                except psutil.Error:
                    pass
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except OSError:
                pass
            p.wait()
            return {"mode": mode, "comm": comm, "error": f"no result within {timeout} s (child job killed)"}
        fo.seek(0)
        fe.seek(0)
        out, err = fo.read(), fe.read()
    lines = [l for l in out.splitlines() if l.startswith("{")]
    if p.returncode != 0 or not lines:
        return {"mode": mode, "error": f"child job exited with status {p.returncode}", "stderr_tail": err[-1500:]}
    try:
        return json.loads(lines[-1])
    except ValueError as e:
        return {"mode": mode, "error": f"unparsable child output: {e!r}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--fib-n", type=int, default=FIB_N)
    ap.add_argument("--cpu-sample-n", type=int, default=FIB_N,
                    help="fibonacci_loop size the CPU oracle proves for cpu_baseline (default: the bench workload itself, ~14 s on 16 threads)")
    ap.add_argument("--cpu-single-n", type=int, default=1000,
                    help="fibonacci_loop size of the single-thread oracle run (default 1000 = BASELINE configs[0]; 0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=min(32, os.cpu_count() or 1),
                    help="OpenMP threads of the multi-thread oracle run: 32 is where its thread-scaling curve peaks on the 256-CPU hosts "
                         "of the pool (profiles/r04a_cpu_scaling.json: 16 -> 2.13e7, 32 -> 2.17e7, 64 -> 1.89e7, 256 -> 2.5e6 cells/s)")
    ap.add_argument("--alt-fib-n", type=int, default=838_000,
                    help="the alternative reading of the metric config (largest column = 2^22 rows), reported as `alt_reading`; 0 = skip")
    ap.add_argument("--alt-steps", type=int, default=3)
    ap.add_argument("--big-fib-n", type=int, default=1_677_000,
                    help="untimed-line leg: fibonacci_loop whose largest LDE column has 2^24 rows (BASELINE configs[3] size on ONE GPU); 0 = skip")
    ap.add_argument("--big-mixed-iters", type=int, default=0,
                    help="untimed-line leg: all-opcodes loop (BASELINE configs[4]); 1_545_000 iterations = 2^26 rows, ~116 GiB of HBM and "
                         "about a minute of host-side input generation: off by default, run by tools/measure_round.sh")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sharded-timeout", type=int, default=300, help="N > 1: time limit in seconds of the sharded-mode child job")
    ap.add_argument("--no-sharded", action="store_true", help="N > 1: skip the second mode (one proof sharded over all ranks)")
    ap.add_argument("--sharded-one-rank-blocks", type=int, default=4,
                    help="N = 1: alternating block pairs of the `sharded_one_rank` leg (cm_prove_sharded with one rank against the single-GPU prover); 0 = skip")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the PCIe-inclusive `end_to_end` measurements")
    ap.add_argument("--no-kprof", action="store_true", help="debug: no in-library HIP-event kernel timing")
    ap.add_argument("--inflight", type=int, default=1,
                    help="independent segment proofs in flight per GPU (one host thread + stream set each)")
    ap.add_argument("--dist-backend", default="nccl", help="debug: gloo lets 2 ranks share one GPU (with --force-device 0)")
    ap.add_argument("--force-device", type=int, default=-1, help="debug: every rank uses this device")
    ap.add_argument("--pipelined", type=int, default=4,
                    help="after the timed region, also measure throughput with this many segment proofs in flight "
                         "(reported as the extra `pipelined` object, N=1 only; 0 = skip)")
    ap.add_argument("--cached-setup-steps", type=int, default=8,
                    help="lone proofs of the untimed `with_cached_setup` leg (preprocessed tree + twiddles kept between proofs); 0 = skip")
    ap.add_argument("--preprocessed-cache", action="store_true",
                    help="NOT the headline number: keep the committed preprocessed tree between proofs (SURVEY 8f-4); the "
                         "default recomputes tree 0 in every proof like the reference does")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: spawn the N ranks ourselves (one process per GPU), exactly the
        # command the driver uses, so that the line printed can never silently be an N=1 measurement.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                 f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # The contract is ONE JSON line on stdout.  Libraries write there too (this image's librccl prints a version banner from its C
    # stdio buffer when its first communicator is created): keep the real stdout for the line and send everything else that is
    # written to file descriptor 1 — by this process or its children — to stderr.
    sys.stdout.flush()
    line_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.force_device >= 0:
            local_rank = args.force_device
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.dist_backend)

    from cairo_m_amd import Backend
    from cairo_m_amd.lib import synth_fibonacci
    be = Backend(local_rank)          # fails loudly without the .so / a GPU
    # the bench's proving threads do nothing but prove: they stay on the GPU's CPUs instead of being moved there and back around
    # every proof (cm_set_cpu_affinity mode 2 = sticky, include/cairom_hip.h; the default, scoped, costs three system calls a proof)
    be.L.cm_set_cpu_affinity(C.c_int32(2))
    be.set_preprocessed_cache(args.preprocessed_cache)
    inp = synth_fibonacci(args.fib_n)  # host: synthetic VM + adapter
    dev = be.upload_input(inp)         # ProverInput resident in HBM before the timed region

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def kprof_report():
        buf = C.create_string_buffer(1 << 16)
        be.L.cm_kprof_report(buf, C.c_size_t(1 << 16))
        return json.loads(buf.value.decode())

    # Warmup: every kernel class is HIP-event timed (per-class table + which class dominates).  Events cost
    # a few microseconds per launch (~2 ms per proof over ~500 launches), so in the timed region only the
    # dominant class keeps its events: `value` is not taxed by the instrumentation, and `roofline.achieved`
    # still comes from launches inside the timed region.
    import threading
    import queue

    class Worker(threading.Thread):
        """One host thread = one proof in flight (the library gives every host thread its own main stream,
        side streams, device pool and pinned upload ring)."""
        def __init__(self):
            super().__init__(daemon=True)
            self.q = queue.Queue()
            self.done = queue.Queue()
            self.start()

        def run(self):
            while True:
                n = self.q.get()
                last = None
                try:
                    for i in range(n):
                        p = be.prove_device(dev)
                        if i == n - 1:
                            last = p.stats()
                        p.free()
                    self.done.put(last)
                except Exception as e:  # noqa: BLE001
                    self.done.put(e)

    workers = [Worker() for _ in range(max(1, args.inflight))]

    def prove_n(n):
        """n proofs spread over the workers; returns the stats of the last proof of worker 0."""
        share = [n // len(workers) + (1 if i < n % len(workers) else 0) for i in range(len(workers))]
        for w, k in zip(workers, share):
            w.q.put(k)
        res = [w.done.get() for w in workers]
        for r in res:
            if isinstance(r, Exception):
                raise r
        return next(r for r in res if r is not None)

    cells = None
    kprof_all = {}
    if args.warmup:
        cells = prove_n(args.warmup * len(workers))["cells"]
    n_prof = 1
    if not args.no_kprof:  # one extra untimed, fully instrumented pass (after the cold-start warmup)
        be.L.cm_kprof_enable(C.c_int32(1))
        workers[0].q.put(1)
        cells = workers[0].done.get()["cells"]
        kprof_all = kprof_report()
    dom_name = max(kprof_all.items(), key=lambda kv: kv[1]["ms"])[0] if kprof_all else None
    be.L.cm_kprof_enable(C.c_int32(0 if args.no_kprof else 1))
    if dom_name:
        be.L.cm_kprof_filter(dom_name.encode())
    sync()
    t0 = time.perf_counter()
    st = prove_n(args.steps)
    cells, phases = st["cells"], st["phase_ms"]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        dist.barrier()
    kprof = kprof_report() if not args.no_kprof else {}
    be.L.cm_kprof_enable(C.c_int32(0))
    pipelined = None
    if world == 1 and args.pipelined > 1:
        # SURVEY §8f-4 segment pipeline: continuation segments are independent proofs, so several can be in
        # flight on one GPU (one host thread each); their launch gaps and host round trips overlap.
        n_pipe = 8 * args.pipelined   # like the streamed legs below: enough proofs that the fill and the drain of the pipeline
                                      # (the last proofs run with fewer neighbours) stay below ~2 % of the figure
        for p in be.prove_many([dev] * args.pipelined, inflight=args.pipelined):   # every worker warms its own pool
            p.free()
        torch.cuda.synchronize()
        tp = time.perf_counter()
        proofs = be.prove_many([dev] * n_pipe, inflight=args.pipelined)
        torch.cuda.synchronize()
        dtp = time.perf_counter() - tp
        for p in proofs:
            p.free()
        pipelined = {"inflight": args.pipelined, "proofs": n_pipe, "api": "cm_prove_many", "ms_per_proof": dtp * 1e3 / n_pipe,
                     "value": n_pipe * cells / dtp, "unit": "M31 trace cells/s",
                     "note": "throughput with several independent segment proofs in flight on the GPU (not the headline value)"}

    cached_setup = None
    if world == 1 and rank == 0 and not args.preprocessed_cache and args.cached_setup_steps > 0:
        # What a service that proves many segments of ONE AIR would run with: the preprocessed tree (tree 0: it depends on the AIR only)
        # and the twiddle tables kept between proofs (cm_set_preprocessed_cache, cm_set_twiddle_cache; proof bytes unchanged:
        # test_preprocessed_cache_keeps_proof_bytes).  NOT the headline: the reference rebuilds both inside prove_cairo_m, and so does
        # every proof of the timed region above.
        be.set_preprocessed_cache(True)
        be.set_twiddle_cache(True)
        for _ in range(4):   # (the pool settles on a new set of blocks once tree 0 stays allocated)
            be.prove_device(dev).free()
        per = []
        for _ in range(args.cached_setup_steps):
            torch.cuda.synchronize()
            tc = time.perf_counter()
            be.prove_device(dev).free()
            torch.cuda.synchronize()
            per.append(time.perf_counter() - tc)
        dtc = sorted(per)[len(per) // 2]   # median of individually timed proofs
        be.set_preprocessed_cache(False)
        be.set_twiddle_cache(False)
        cached_setup = {"ms_per_proof": dtc * 1e3, "value": cells / dtc, "unit": "M31 trace cells/s", "proofs_timed": args.cached_setup_steps,
                        "statistic": "median of individually timed lone proofs",
                        "note": "lone proofs with the preprocessed tree and the twiddle tables cached between proofs (same proof bytes); "
                                "not the headline value: the timed region rebuilds both in every proof like the reference does"}

    alt = None
    if world == 1 and rank == 0 and args.alt_fib_n > 0:
        # the other reading of "2^22 rows" — the LARGEST COLUMN has 2^22 rows (n = 838 000, store_fp_imm 3.35 M rows)
        a_inp = synth_fibonacci(args.alt_fib_n)
        a_dev = be.upload_input(a_inp)
        be.prove_device(a_dev).free()
        torch.cuda.synchronize()
        ta = time.perf_counter()
        for _ in range(args.alt_steps):
            pa = be.prove_device(a_dev)
            a_stats = pa.stats()
            pa.free()
        torch.cuda.synchronize()
        a_dt = (time.perf_counter() - ta) / args.alt_steps
        alt = {"workload": f"fibonacci_loop n={args.alt_fib_n} ({a_inp.steps} VM steps, one segment, largest column 2^"
                           f"{max(be.component_log_size(a_dev, c) for c in range(34))} rows)",
               "cells": a_stats["cells"], "ms_per_proof": a_dt * 1e3, "value": a_stats["cells"] / a_dt, "unit": "M31 trace cells/s",
               "steps_per_s": a_inp.steps / a_dt, "proofs_timed": args.alt_steps}
        be.free_input(a_dev)
        a_inp.free()
        be.pool_trim()

    big_legs = []
    if world == 1 and rank == 0:
        # whole-path model at the larger sizes north_star names (same 52 B / cell model): the share of protocol-serial latency
        # (17 tree tops, 12 FRI layers, transcript steps) shrinks with the size, so the fraction of the HBM roofline rises
        def leg(name, make_input, reps, upload=None):
            try:
                b_inp = make_input()
                b_dev = (upload or be.upload_input)(b_inp)
                for _ in range(2):   # two untimed proofs: the device pool settles on its blocks for this size in the second one (a
                    be.prove_device(b_dev).free()   # thread parks part of a proof's teardown until its next proof: DESIGN §3.1)
                torch.cuda.synchronize()
                tb = time.perf_counter()
                for _ in range(reps):
                    pb = be.prove_device(b_dev)
                    b_stats = pb.stats()
                    pb.free()
                torch.cuda.synchronize()
                b_dt = (time.perf_counter() - tb) / reps
                big_legs.append({"workload": name, "vm_steps": b_stats["steps"], "cells": b_stats["cells"], "ms_per_proof": b_dt * 1e3,
                                 "value": b_stats["cells"] / b_dt, "unit": "M31 trace cells/s",
                                 "largest_column_log2": max(be.component_log_size(b_dev, c) for c in range(34)),
                                 "whole_path_model": {"bytes_per_cell": MODEL_BYTES_PER_CELL,
                                                      "achieved_GBs": MODEL_BYTES_PER_CELL * b_stats["cells"] / b_dt / 1e9,
                                                      "frac": MODEL_BYTES_PER_CELL * b_stats["cells"] / b_dt / 1e9 / HBM_PEAK_GBS},
                                 "proofs_timed": reps})
                be.free_input(b_dev)
                b_inp.free()
                be.pool_trim()
            except Exception as e:  # noqa: BLE001  (a leg that does not fit costs only its own entry)
                big_legs.append({"workload": name, "error": str(e)[:200]})
        if args.big_fib_n > 0:
            leg(f"fibonacci_loop n={args.big_fib_n} (one segment, 2^24-row LDE columns; BASELINE configs[3] size on one GPU)",
                lambda: synth_fibonacci(args.big_fib_n), 2)
        if args.big_mixed_iters > 0:
            from cairo_m_amd.lib import vm_segment
            from cairo_m_amd.workloads import all_opcodes_program
            leg(f"all-opcodes loop, {args.big_mixed_iters} iterations (BASELINE configs[4] on one GPU; runner segment -> device adapter)",
                lambda: vm_segment(all_opcodes_program(args.big_mixed_iters)[0], entry_pc=0, args=(), n_returns=0), 1, upload=be.adapt_segment)

    sharded = None
    if world > 1 and not args.no_sharded:
        # Second mode (BASELINE configs[3]): the `world` ranks prove ONE segment together — components split across
        # the ranks, rows split for Merkle hashing and DEEP quotients, collectives over RCCL (cairo_m_amd/sharded.py).  This is
        # single-proof LATENCY scaling (strong scaling); the headline `value` above stays the replica throughput.
        # It runs as a CHILD job (its own N ranks, launched by rank 0 with a time limit) after the replica measurement is
        # complete: a collective that hangs or a rank that dies in this mode can then cost the `sharded` object, never the line.
        store = dist.distributed_c10d._get_default_store()
        if rank == 0:
            sharded = run_sharded_child(args, world)
            if sharded and "error" not in sharded and args.dist_backend == "nccl":
                # the same proof with the in-library RCCL communicator (no host sync, no Python in the data path); a failure
                # here costs only this sub-object
                sharded["in_library_rccl"] = run_sharded_child(args, world, comm="rccl", timeout=min(180, args.sharded_timeout))
            store.set("cm_sharded_child_done", "1")
        else:
            store.wait(["cm_sharded_child_done"], datetime.timedelta(seconds=2 * args.sharded_timeout + 120))
    verified = None
    hip_words = None
    end_to_end = None
    if rank == 0:
        # one more (untimed) proof of the same input: the product verifier must accept it, and its words are what the
        # cpu_baseline leg compares the oracle's proof with
        p = be.prove_device(dev)
        hip_words = p.words().copy()
        vrc, verr = p.verify()
        verified = vrc == 0
        p.free()
        if not verified:
            sys.exit(f"bench.py: the HIP proof of the bench workload does not verify: {verr}")
    sharded_one = None
    if rank == 0 and world == 1 and args.sharded_one_rank_blocks > 0:
        # SURVEY 8e-2 on the one GPU there is: the SAME segment through cm_prove_sharded with ONE rank (the in-library stream-ordered
        # RCCL communicator, world = 1) in blocks alternating with the single-GPU prover inside this process — what the sharded
        # path costs by itself (collective plumbing, sub-root gathers, host-planned decommitment), before any rank is added.
        # A failure here costs only this object.
        try:
            import numpy as np
            from cairo_m_amd.sharded import RcclComm, prove_sharded, shard_plan
            _, s_words = shard_plan(inp, 1, be.L, None)
            idb = (C.c_uint8 * 128)()
            be._ck(be.L.cm_rccl_unique_id(idb))
            comm = RcclComm(be, s_words, rank=0, world=1, id_bytes=bytes(idb))
            ps = prove_sharded(be, dev, comm)
            s_same = bool(ps.words().size == hip_words.size and np.array_equal(ps.words(), hip_words))
            ps.free()

            def s_block(sh, n=4):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    (prove_sharded(be, dev, comm) if sh else be.prove_device(dev)).free()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / n * 1e3
            for _ in range(2):
                s_block(True); s_block(False)
            t_sh, t_one = [], []
            for r in range(args.sharded_one_rank_blocks):
                for sh in ((False, True) if r % 2 == 0 else (True, False)):
                    s_block(sh, 2)   # untimed after every switch
                    (t_sh if sh else t_one).append(s_block(sh))
            ps = prove_sharded(be, dev, comm)   # (a warm proof's GPU-side phase times)
            s_phases = {k: round(v, 3) for k, v in ps.stats()["phase_ms"].items()}
            ps.free()
            comm.free()
            be.pool_trim()
            sharded_one = {"ms_per_proof": statistics.median(t_sh), "single_gpu_ms_same_process": statistics.median(t_one),
                           "paired_difference_ms": statistics.median([x - y for x, y in zip(t_sh, t_one)]),
                           "blocks": args.sharded_one_rank_blocks, "proofs_per_block": 4, "world": 1, "comm": "in-library RCCL (stream-ordered)",
                           "bit_identical_to_single_gpu_proof": s_same, "phase_ms": s_phases,
                           "note": "cm_prove_sharded with one rank against cm_prove_device in alternating blocks of lone proofs inside this "
                                   "process: the sharded path's own overhead (not a scaling figure; the multi-rank figures are the `sharded` "
                                   "object of a --gpus N run)"}
        except Exception as e:   # noqa: BLE001 — the bench line must survive
            sharded_one = {"error": repr(e)[:300]}
    if rank == 0 and world == 1 and not args.no_end_to_end:
        # PCIe-inclusive figures (never `value`): (a) cm_prove_segment from a host ProverInput = upload + prove;
        # (b) runner segment (trace + memory log in host memory) -> device adapter -> prove
        from cairo_m_amd.lib import synth_fibonacci_segment
        def best(f, reps=3):
            ts = []
            for _ in range(reps):
                torch.cuda.synchronize()
                t = time.perf_counter()
                f()
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t) * 1e3)
            return min(ts), sum(ts) / len(ts)
        def from_host():
            be.prove(inp).free()
        seg = synth_fibonacci_segment(args.fib_n)
        def from_runner():
            d = be.adapt_segment(seg)
            be.prove_device(d).free()
            be.free_input(d)
        h_min, h_avg = best(from_host)
        r_min, r_avg = best(from_runner)
        # streaming ingest (cm_prove_many_host / cm_prove_many_segments): the same host inputs, uploads / device adapter of item
        # i + 1 under the proofs of the items before it — compare with `pipelined.ms_per_proof` (inputs resident in HBM)
        streamed = None
        if args.pipelined > 1:
            n_str = 8 * args.pipelined   # a STREAM of segments: the first proof cannot start before the first upload has landed
                                         # (~6 ms of pipeline fill, once), so the run is long enough to amortise it below 2 %
            def run_streamed(fn, item):
                for p in fn([item] * args.pipelined, inflight=args.pipelined):
                    p.free()
                torch.cuda.synchronize()
                t = time.perf_counter()
                ps = fn([item] * n_str, inflight=args.pipelined)
                torch.cuda.synchronize()
                d = (time.perf_counter() - t) * 1e3 / n_str
                for p in ps:
                    p.free()
                return d
            streamed = {"inflight": args.pipelined, "proofs": n_str,
                        "pipelined_from_host_ms_per_proof": run_streamed(be.prove_many_host, inp),
                        "pipelined_from_segments_ms_per_proof": run_streamed(be.prove_many_segments, seg),
                        "api": "cm_prove_many_host / cm_prove_many_segments"}
        seg.free()
        end_to_end = {"host_prover_input_ms": {"min": h_min, "avg": h_avg, "api": "cm_prove_segment (pageable host ProverInput: upload + prove)"},
                      "runner_segment_ms": {"min": r_min, "avg": r_avg, "api": "cm_adapt_segment_device + cm_prove_device (runner trace + memory log in host memory -> device adapter -> proof)"},
                      "cells_per_s_from_host_input": cells / (h_min * 1e-3),
                      **(streamed or {}),
                      **({"streamed_vs_resident": streamed["pipelined_from_host_ms_per_proof"] / pipelined["ms_per_proof"]}
                         if streamed and pipelined else {}),
                      "note": "PCIe-inclusive; never the headline `value` (inputs resident in HBM)"}

    if rank == 0:
        ms_per_step = dt * 1e3 / args.steps
        value = world * args.steps * cells / dt
        dom = max(kprof.items(), key=lambda kv: kv[1]["ms"]) if kprof else None
        roofline = None
        if dom:
            name, k = dom
            achieved = k["bytes"] / (k["ms"] * 1e-3) / 1e9
            traffic, traffic_src = pmc_traffic(name)
            # `bound`: what limits the class.  The Blake2s and butterfly classes are integer-VALU-issue-bound (DESIGN.md §3,
            # profiles/*_pmc_sq.json: the SIMDs' VALU is saturated while HBM runs at a quarter of its peak) — "valu", with the
            # HBM figures (achieved / peak / frac: algorithmic bytes over the HIP-event duration) kept beside it and the
            # integer-ALU fraction in `alu`
            roofline = {"bound": "valu" if name in ALU_PEAK else "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                        "avg_launch_ms": k["ms"] / k["calls"], "launches": k["calls"],
                        "alu": ({"unit": ALU_PEAK[name][1], "achieved": k.get("work", 0.0) / (k["ms"] * 1e-3),
                                 "peak": ALU_PEAK[name][0], "frac": k.get("work", 0.0) / (k["ms"] * 1e-3) / ALU_PEAK[name][0],
                                 **({"register_only_rate": B2S_REGISTER_ONLY,
                                     "frac_of_register_only": k.get("work", 0.0) / (k["ms"] * 1e-3) / B2S_REGISTER_ONLY,
                                     "register_only_note": "the same compression code looping on registers only, whole GPU (tools/chain_lab.hip, "
                                                           "profiles/r03w_chain_lab.txt): the rate no Merkle kernel can exceed with this op mix"}
                                    if name == "k_merkle_layer" else {}),
                                 **({"alone": {"achieved": al["work"] / (al["ms"] * 1e-3), "frac_of_peak": al["work"] / (al["ms"] * 1e-3) / ALU_PEAK[name][0],
                                               "frac_of_register_only": al["work"] / (al["ms"] * 1e-3) / B2S_REGISTER_ONLY,
                                               "launches": al["calls"], "ms_per_step": al["ms"] / n_prof, "compressions_per_step": al["work"] / n_prof,
                                               "note": "the same class over the launches that have the GPU to themselves (FRI commit phase and composition tree "
                                                       "of a lone proof: one kernel at a time on the main stream; instrumented pass).  `achieved` above is over "
                                                       "ALL launches of the timed region, including the trace / interaction trees whose hashing the commitment "
                                                       "pipeline overlaps with the transforms of the next size group — both kernels are slower while they share "
                                                       "the GPU, the sum is what the pipeline shortens"}}
                                    if name == "k_merkle_layer" and (al := kprof_all.get("k_merkle_layer(alone)")) and al.get("ms") else {}),
                                 "note": "this class is integer-VALU-bound, not HBM-bound (DESIGN.md §3); peak = 1 / sum(ops_i / measured lane-op rate_i), "
                                         "rates from profiles/r03k_valu_lab.txt (tools/valu_lab.hip)"}
                                if name in ALU_PEAK and k.get("work") else None),
                        "algorithmic_bytes_per_launch": k["bytes"] / k["calls"],
                        "whole_path_model": {"bytes_per_cell": MODEL_BYTES_PER_CELL,
                                             "achieved_GBs": MODEL_BYTES_PER_CELL * cells / (ms_per_step * 1e-3) / 1e9,
                                             "frac": MODEL_BYTES_PER_CELL * cells / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
                        "valu_model": valu_model(ms_per_step),
                        "whole_path_model_larger_sizes": big_legs or None,
                        "stages": stage_roofline(kprof_all, n_prof, cells),
                        "stages_note": "SURVEY §8d per-stage check: bytes per cell x committed cells / summed HIP-event intervals of the "
                                       "stage's kernel classes in the instrumented pass.  With the commitment pipeline the transform and "
                                       "Merkle intervals of a tree overlap in time (each includes the time it shared the GPU with the "
                                       "other), so their sum exceeds the commit phases' wall time",
                        "kernels_note": "per-class HIP-event totals from the instrumented (untimed) pass after the warmup; '(region)' = "
                                        "one interval around a fork/join of per-component launches on side streams",
                        "kernels": {n: {"ms_per_step": v["ms"] / n_prof, "launches_per_step": v["calls"] / n_prof,
                                        "GBs": v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] else None}
                                    for n, v in kprof_all.items()}}
        # the reference's own speed figure (prover.rs:133-138): 2^trace_log_size / duration of Stwo's `prove` call, which covers
        # constraint evaluation ... decommitment (everything after the three trace commitments)
        trace_log = max(20, max(be.component_log_size(dev, c) for c in range(34)))
        stark_ms = sum(phases[k] for k in ("constraints", "composition_commit", "oods_sampling", "quotients", "fri_commit", "pow", "decommit"))
        out = {"metric": "M31 trace cells/sec proved, fibonacci_loop 2^22 rows; end-to-end proof ms",
               "value": value, "unit": "M31 trace cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "u32", "data": "synthetic",
               "config": {"workload": f"fibonacci_loop n={args.fib_n} ({inp.steps} VM steps, one segment, "
                                      f"{cells} committed trace cells), REGULAR_96_BITS PCS config, "
                                      "ProverInput resident in HBM (upload excluded: see `end_to_end`), twiddle tables "
                                      "and the preprocessed tree rebuilt inside every proof like the reference does"
                                      + (", preprocessed tree cached between proofs" if args.preprocessed_cache else ""),
                          "cells_per_proof": cells, "cpu_affinity": "sticky (cm_set_cpu_affinity(2))",
                          "vm_steps": inp.steps, "parallelism": f"{world} independent segment replica(s) x {len(workers)} proof(s) in flight per GPU"},
               "steps_per_s": world * args.steps * inp.steps / dt,
               "mhz": {"value": (1 << trace_log) / (stark_ms * 1e-3) / 1e6, "trace_log_size": trace_log, "stark_prove_ms": stark_ms,
                       "note": "the reference's own figure, prover.rs:133-138: 2^trace_log_size / duration of the Stwo `prove` call "
                               "(constraints ... decommit phases of the last timed proof)"},
               "gpu_idle_traced": traced_idle(),
               "alt_reading": alt,
               "phase_ms": phases, "roofline": roofline, "pipelined": pipelined, "with_cached_setup": cached_setup, "sharded": sharded, "sharded_one_rank": sharded_one, "end_to_end": end_to_end,
               "proof_verified": verified}
        if world == 1 and not args.no_cpu_baseline:
            same = args.cpu_sample_n == args.fib_n
            parity, out["cpu_baseline"] = cpu_baseline(args.cpu_sample_n, args.cpu_single_n, args.cpu_threads, hip_words if same else None)
            # bit-exactness AT the metric config: the oracle proves the very ProverInput the GPU was timed on
            out["parity_at_metric_config"] = parity
            sc = latest_profile("cpu_scaling.json") or os.path.join(ROOT, "profiles", "r04_cpu_scaling.json")
            if os.path.exists(sc):
                try:
                    out["cpu_baseline"]["thread_scaling"] = dict(json.load(open(sc)), source=os.path.relpath(sc, ROOT))
                except (OSError, ValueError):
                    pass
            if parity is False:
                print(json.dumps(out), file=line_out, flush=True)
                sys.exit("bench.py: the HIP proof differs from the CPU oracle's proof of the same input")
        else:
            out["cpu_baseline"] = None
            out["parity_at_metric_config"] = None
        print(json.dumps(out), file=line_out, flush=True)
    be.free_input(dev)
    inp.free()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
