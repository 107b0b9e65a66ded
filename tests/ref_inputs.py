"""Inputs of the reference's own prover tests (crates/prover/tests/prover.rs), restated for both boxes.

* unchanged_memory_input(): `test_prove_and_verify_unchanged_memory` (prover.rs:33-112) builds its ProverInput BY HAND — four
  memory cells that never change, no instruction at all, empty public ranges — so it can be reconstructed here exactly, with
  no compiler and no runner.  (The reference iterates its HashMap for the memory rows, components/memory.rs:105-109; this
  build fixes ascending addresses, and a reference-produced golden carries the order it used.)
* the programs below are hand-assembled CASM with the structure of the reference's fixtures (test_data/functions/*.cm): the
  compiler is not available, so they are NOT claimed to be byte-identical to its output.
"""
import numpy as np

P = 2**31 - 1


def unchanged_memory_arrays():
    from cairo_m_amd.lib import partial_merkle_tree
    cells = [(0, 1, 2, 3, 4), (1, 5, 6, 7, 8), (2, 9, 10, 11, 12), (3, 13, 14, 15, 16)]   # prover.rs:34-51
    init_tree, init_root = partial_merkle_tree(cells, True)      # PublicAddressRanges::default(): nothing public
    fin_tree, fin_root = partial_merkle_tree(cells, False)
    mem = [[a, v0, v1, v2, v3, 0, 0] for a, v0, v1, v2, v3 in cells]   # (address, value, clock 0, multiplicity 0)
    return {"regs": [0, 0, 0, 0], "initial_memory": mem, "final_memory": mem, "initial_tree": init_tree, "final_tree": fin_tree,
            "roots": [init_root, fin_root], "ranges": [0, 0, 0, 0, 0, 0]}


def unchanged_memory_input():
    from cairo_m_amd.lib import ArrayInput
    return ArrayInput(unchanged_memory_arrays())


def recursive_fib_program():
    """fib(n) = n < 2 ? n : fib(n-1) + fib(n-2), the shape of test_data/functions/fibonacci.cm (prover.rs:175-200): deep
    call_abs_imm / ret.  Frame of a call (runner/src/vm/instructions/call.rs:6-31): [fp-4] = n, [fp-3] = return slot,
    [fp-2] = old fp, [fp-1] = return pc; `call abs frame_off target` stores fp / return pc at fp+frame_off / +1 and enters
    the callee with fp + frame_off + 2.  Words are [opcode, operands...]; m(-k) is P - k."""
    m = lambda k: (P + k) % P
    return [
        [14, m(-4), 2],          # 0: jnz [fp-4] -> 2              (n != 0)
        [13, 10],                # 1: jmp rel -> 11                (n == 0: return n)
        [4, m(-4), m(-1), 0],    # 2: [fp+0] = n - 1
        [14, 0, 2],              # 3: jnz [fp+0] -> 5              (n != 1)
        [13, 7],                 # 4: jmp rel -> 11                (n == 1: return n)
        [4, 0, 0, 1],            # 5: [fp+1] = n - 1               (argument of the first call)
        [10, 3, 0],              # 6: call abs, frame_off 3: callee fp = fp + 5, its n at fp+1, its return slot at fp+2
        [4, m(-4), m(-2), 6],    # 7: [fp+6] = n - 2               (argument of the second call)
        [10, 8, 0],              # 8: call abs, frame_off 8: callee fp = fp + 10, its n at fp+6, its return slot at fp+7
        [0, 2, 7, m(-3)],        # 9: return slot = fib(n-1) + fib(n-2)
        [11],                    # 10: ret
        [4, m(-4), 0, m(-3)],    # 11: return slot = n
        [11],                    # 12: ret
    ]
