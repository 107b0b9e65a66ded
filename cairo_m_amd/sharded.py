"""Host side of the intra-proof sharded prover (include/cairom_hip.h: cm_comm, cm_shard_plan, cm_prove_sharded).

The library does the proving and the packing; THIS module provides the two collectives it asks for, over
torch.distributed: backend "nccl" (= RCCL over xGMI on a multi-GPU node: device tensors go straight into
all_to_all_single / all_gather_into_tensor) or "gloo" (tests: several ranks sharing one GPU, or no GPU at all — the
staging buffers are copied through host memory).  One process per rank; launch with

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 -m cairo_m_amd.sharded --fib-n 419000

Every rank ends up with the same proof, bit-identical to the single-GPU one (tests/test_gpu_sharded.py).
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

from .lib import N_COMPONENTS, Backend, CmError, Proof, _cfg, load_library, synth_fibonacci

_A2A = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64))
_AG = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_uint64)


_SETSTREAM = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_uint64)
_ABORT = C.CFUNCTYPE(None, C.c_void_p)


class CmComm(C.Structure):
    _fields_ = [("struct_size", C.c_uint32),                             # sizeof(cm_comm) of THIS binding: the library reads no further
                ("rank", C.c_uint32), ("world", C.c_uint32), ("ctx", C.c_void_p), ("send_buf", C.c_void_p), ("recv_buf", C.c_void_p),
                ("buf_words", C.c_uint64), ("all_to_all_v", _A2A), ("all_gather", _AG),
                ("flags", C.c_uint32), ("set_stream", _SETSTREAM),       # 0 / NULL: the blocking form (this module's TorchComm)
                ("abort", _ABORT)]                                       # NULL: the process group's own timeout releases the peers

    def __init__(self, *fields, **kw):   # CmComm(rank, world, ctx, ...): struct_size is filled in here
        super().__init__(C.sizeof(CmComm), *fields, **kw)


def _cfg_arg(cfg):
    return (C.c_uint32 * 4)(*cfg) if cfg else None


def shard_plan(host_input, world, lib=None, cfg=None):
    """(owner per component, staging words) under PCS config `cfg` — host code, identical on every rank."""
    L = lib or load_library()
    owner = (C.c_int32 * N_COMPONENTS)()
    words = C.c_uint64(0)
    rc = L.cm_shard_plan(host_input.view, _cfg_arg(cfg), C.c_uint32(world), owner, C.byref(words))
    if rc:
        raise CmError(f"cm_shard_plan failed with status {rc}")
    return list(owner), words.value


def shard_plan_columns(host_input, world, lib=None, cfg=None):
    """column-level plan (cm_shard_plan_columns): (owner of every trace column, owner of every interaction column, cells per rank)"""
    L = lib or load_library()
    ntr, nit = C.c_uint32(0), C.c_uint32(0)
    load = (C.c_uint64 * 8)()
    rc = L.cm_shard_plan_columns(host_input.view, _cfg_arg(cfg), C.c_uint32(world), None, C.byref(ntr), None, C.byref(nit), load)   # the counts
    if rc:
        raise CmError(f"cm_shard_plan_columns failed with status {rc}")
    tr, it = (C.c_int32 * max(1, ntr.value))(), (C.c_int32 * max(1, nit.value))()
    rc = L.cm_shard_plan_columns(host_input.view, _cfg_arg(cfg), C.c_uint32(world), tr, C.byref(ntr), it, C.byref(nit), load)
    if rc:
        raise CmError(f"cm_shard_plan_columns failed with status {rc}")
    return list(tr[:ntr.value]), list(it[:nit.value]), list(load[:world])


class TorchComm:
    """cm_comm over torch.distributed.  `staging` = (send, recv) int32 tensors: CUDA tensors for a real run, CPU tensors when
    the collectives are exercised without a GPU (tests/test_multirank_cpu.py)."""

    def __init__(self, staging_words, device=None, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        dev = torch.device("cuda", device) if device is not None else torch.device("cpu")
        self.send = torch.empty(staging_words, dtype=torch.int32, device=dev)
        self.recv = torch.empty(staging_words, dtype=torch.int32, device=dev)
        self.direct = self.backend == "nccl"        # device tensors straight into the collective
        self.bytes_moved = 0
        self.calls = 0
        self._a2a, self._ag = _A2A(self._all_to_all_v), _AG(self._all_gather)   # keep the thunks alive
        self.c = CmComm(self.rank, self.world, None, self.send.data_ptr(), self.recv.data_ptr(), staging_words, self._a2a, self._ag,
                        0, _SETSTREAM())

    def _sync(self):
        if self.send.is_cuda:
            self.torch.cuda.synchronize()

    def _all_to_all_v(self, _ctx, send_words, recv_words):
        try:
            n = self.world
            sw = [int(send_words[i]) for i in range(n)]
            rw = [int(recv_words[i]) for i in range(n)]
            src = self.send[:sum(sw)]
            dst = self.recv[:sum(rw)]
            if self.direct:
                self.dist.all_to_all_single(dst, src, output_split_sizes=rw, input_split_sizes=sw, group=self.group)
            else:
                s = src.cpu()
                r = self.torch.empty(sum(rw), dtype=self.torch.int32)
                outs = list(r.split(rw)) if sum(rw) else [r[:0]] * n
                ins = list(s.split(sw)) if sum(sw) else [s[:0]] * n
                # gloo has no all_to_all on every build: pairwise exchange, deadlock-free order
                reqs = []
                for k in range(n):
                    peer_s, peer_r = (self.rank + k) % n, (self.rank - k) % n
                    if k == 0:
                        outs[self.rank].copy_(ins[self.rank])
                        continue
                    if sw[peer_s]:
                        reqs.append(self.dist.isend(ins[peer_s].contiguous(), self._global(peer_s), group=self.group))
                    if rw[peer_r]:
                        reqs.append(self.dist.irecv(outs[peer_r], self._global(peer_r), group=self.group))
                for q in reqs:
                    q.wait()
                dst.copy_(r)
            self._sync()
            self.bytes_moved += 4 * sum(sw)
            self.calls += 1
            return 0
        except Exception as e:  # noqa: BLE001 — a Python exception must not unwind through the C frame
            print(f"[rank {self.rank}] all_to_all_v failed: {e!r}", file=sys.stderr)
            return 1

    def _global(self, group_rank):
        return self.dist.get_global_rank(self.group, group_rank) if self.group is not None else group_rank

    def _all_gather(self, _ctx, words_per_rank):
        try:
            w, n = int(words_per_rank), self.world
            if w == 0:
                return 0
            src, dst = self.send[:w], self.recv[:w * n]
            if self.direct:
                self.dist.all_gather_into_tensor(dst, src, group=self.group)
            else:
                s = src.cpu()
                parts = [self.torch.empty(w, dtype=self.torch.int32) for _ in range(n)]
                self.dist.all_gather(parts, s, group=self.group)
                dst.copy_(self.torch.cat(parts))
            self._sync()
            self.bytes_moved += 4 * w * (n - 1)
            self.calls += 1
            return 0
        except Exception as e:  # noqa: BLE001
            print(f"[rank {self.rank}] all_gather failed: {e!r}", file=sys.stderr)
            return 1


class RcclComm:
    """The IN-LIBRARY communicator (include/cairom_hip.h cm_rccl_*): RCCL collectives enqueued on the prover's own stream, no
    host synchronisation and no Python in the data path.  torch.distributed is used ONCE, to hand rank 0's 128-byte RCCL id
    to the other ranks (control plane); pass `id_bytes` to skip even that (env / file / MPI launchers)."""

    def __init__(self, backend, staging_words, rank=None, world=None, id_bytes=None):
        self.L = backend.L
        if id_bytes is None:
            import torch.distributed as dist
            rank, world = dist.get_rank(), dist.get_world_size()
            box = [None]
            if rank == 0:
                buf = (C.c_uint8 * 128)()
                backend._ck(self.L.cm_rccl_unique_id(buf))
                box[0] = bytes(buf)
            dist.broadcast_object_list(box, src=0)
            id_bytes = box[0]
        self.rank, self.world = rank, world
        self.h = C.c_void_p()
        backend._ck(self.L.cm_rccl_comm_create((C.c_uint8 * 128)(*id_bytes), C.c_uint32(rank), C.c_uint32(world),
                                                C.c_uint64(staging_words), C.byref(self.h)))
        self.L.cm_rccl_comm_view.restype = C.POINTER(CmComm)
        self.c = self.L.cm_rccl_comm_view(self.h).contents
        self.calls = 0
        self.bytes_moved = 0      # (not counted on this path: nothing passes through Python)

    def free(self):
        if self.h:
            self.L.cm_rccl_comm_destroy(self.h)
            self.h = None


def prove_sharded(backend, dev_input, comm, cfg=None):
    """cm_prove_sharded: every rank of comm's group calls this with the SAME input; returns this rank's copy of the proof."""
    h = C.c_void_p()
    backend._ck(backend.L.cm_prove_sharded(dev_input, _cfg(cfg), C.byref(comm.c), C.byref(h)))
    return Proof(backend.L, h)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fib-n", type=int, default=1000)
    ap.add_argument("--mixed-iters", type=int, default=0,
                    help="prove the all-opcode loop (cairo_m_amd/workloads.py, BASELINE configs[4]) with this many iterations instead")
    ap.add_argument("--dist-backend", default="nccl")
    ap.add_argument("--comm", default="torch", choices=["torch", "rccl"],
                    help="torch: collectives through torch.distributed callbacks (blocking); rccl: the library's own stream-ordered "
                         "RCCL communicator (cm_rccl_comm_create) — needs one GPU per rank")
    ap.add_argument("--force-device", type=int, default=-1, help="every rank uses this GPU (tests: ranks sharing one GPU over gloo)")
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--cfg", default="", help="PcsConfig as pow_bits,log_blowup_factor,log_last_layer_degree_bound,n_queries (default REGULAR_96_BITS)")
    ap.add_argument("--out", default="", help="rank r writes its proof words to <out>.<r>.npy")
    ap.add_argument("--check-single", action="store_true", help="every rank also makes the single-GPU proof and compares all words")
    ap.add_argument("--json", action="store_true", help="rank 0 prints one JSON line (bench.py's `sharded` object) instead of a dict")
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    local = int(os.environ.get("LOCAL_RANK", "0")) if a.force_device < 0 else a.force_device
    torch.cuda.set_device(local)
    if a.dist_backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(a.dist_backend)
    be = Backend(local)
    if a.mixed_iters:
        from .lib import vm_run
        from .workloads import all_opcodes_program
        inp = vm_run(all_opcodes_program(a.mixed_iters)[0], entry_pc=0, args=(), n_returns=0)
    else:
        inp = synth_fibonacci(a.fib_n)
    cfg = tuple(int(x) for x in a.cfg.split(",")) if a.cfg else None
    owner, words = shard_plan(inp, dist.get_world_size(), be.L, cfg)   # the plan and the staging bound of THIS config
    comm = RcclComm(be, words) if a.comm == "rccl" else TorchComm(words, device=local)
    dev = be.upload_input(inp)
    p = prove_sharded(be, dev, comm, cfg)     # warm-up + the proof that is written out
    w = p.words().copy()
    cells = p.stats()["cells"]
    p.free()
    same = None
    if a.check_single:
        q = be.prove_device(dev, cfg)
        same = bool(q.words().size == w.size and (q.words() == w).all())
        q.free()
    ms, phases = [], {}
    for _ in range(a.steps):
        dist.barrier()
        torch.cuda.synchronize()
        t = time.perf_counter()
        q = prove_sharded(be, dev, comm, cfg)
        torch.cuda.synchronize()
        ms.append((time.perf_counter() - t) * 1e3)
        phases = {k: round(v, 3) for k, v in q.stats()["phase_ms"].items()}
        q.free()
    if a.out:
        np.save(f"{a.out}.{dist.get_rank()}.npy", w)
    on_gpu = a.dist_backend == "nccl"
    tt = torch.tensor([sum(ms), 0.0 if same in (None, True) else 1.0], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    if a.json:
        if dist.get_rank() == 0:
            import json
            total = float(tt[0].item())
            print(json.dumps({
                "mode": "one proof sharded over all ranks (strong scaling)", "world": dist.get_world_size(),
                "ms_per_proof": total / max(1, a.steps), "value": cells * a.steps / (total * 1e-3) if total else None,
                "unit": "M31 trace cells/s", "bit_identical_to_single_gpu_proof": (tt[1].item() == 0.0) if a.check_single else None,
                "component_owner": owner, "collectives_per_proof": comm.calls // (a.steps + 1),
                "MB_sent_per_rank_per_proof": comm.bytes_moved / (a.steps + 1) / 1e6, "phase_ms": phases,
                "note": "components are the sharding unit, large opcode components split by rows (generation, lookups, constraints) and "
                        "columns (transforms); Merkle hashing of all four trees, the DEEP quotients and the FRI "
                        "layers above 2^16 rows are row-sharded; the (cheap) transforms of trees 0 / 3 and the small FRI layers are "
                        "replicated; time = max over ranks of the wall time of `steps` proofs"}))
    elif dist.get_rank() == 0:
        print({"world": dist.get_world_size(), "owner": owner, "staging_words": words, "ms": ms, "phase_ms": phases, "cells": cells,
               "comm_calls_per_proof": comm.calls // (a.steps + 1), "comm_MB_per_proof": comm.bytes_moved / (a.steps + 1) / 1e6})
    be.free_input(dev)
    inp.free()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
