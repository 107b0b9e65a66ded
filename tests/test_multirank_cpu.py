"""N > 1 path on CPU (gloo, world_size 2): the bench's sharding logic — every rank proves an independent
segment (no data-path collective); the only collectives are the barrier and the max-over-ranks time.
Here each rank checks its own segment with the CPU oracle and the ranks agree on the aggregate."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, time
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from cairo_m_amd.lib import vm_run
    from tests.oracle_binding import Oracle
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    # a cell cannot be read and written in the same step (clock deltas must be >= 1): ping-pong between two cells
    prog = [[9, 1, 0]] + [[4, 0, 1, 1], [4, 1, 1, 0]] * 3 + [[4, 0, 1, 1]] + [[11]]
    # one program cut into `world` continuation segments: rank r owns segment r
    inp = vm_run(prog, max_steps=5, segment=rank)
    assert inp.n_segments == world, inp.n_segments
    orc = Oracle(os.path.join(%r, "oracle", "liboracle.so"))
    dist.barrier()
    t0 = time.perf_counter()
    rc, err = orc.assert_constraints(inp.view)
    dt = time.perf_counter() - t0
    assert rc == 0, err
    t = torch.tensor([dt, float(inp.steps)], dtype=torch.float64)
    tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
    assert tmax[0] >= t[0]
    assert int(tsum[1]) == 9  # 9 VM steps in total across the segments
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "rank%%d.ok" %% rank), "w").write("ok")
""")


def test_two_rank_gloo(tmp_path, oracle):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % (ROOT, ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29517", str(script)],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert (tmp_path / "rank0.ok").exists() and (tmp_path / "rank1.ok").exists()
