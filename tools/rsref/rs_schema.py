#!/usr/bin/env python3
"""Serde field order of the Cairo-M-owned structs inside `Proof<H>` (crates/prover/src/lib.rs:61-73), extracted from the
reference's Rust source AT GENERATION TIME -> tests/golden/proof_schema.json (names and order only: data, no source text).

  Proof            lib.rs:61-73
  Claim            components/mod.rs:28-39          InteractionClaim  components/mod.rs:65-76
  opcodes::Claim   fields = module names of `define_opcodes!` in macro order (components/opcodes/mod.rs:223-268)
  <c>::Claim       { log_size }                     <c>::InteractionClaim { claimed_sum }
  PublicData       public_data.rs:212-227           PublicEntries public_data.rs:58-63     VmRegisters = cairo_m_common::State, common/src/state.rs:10-13

`stark_proof` is Stwo's `StarkProof<H>` (CommitmentSchemeProof / FriProof / ...): not vendored, so its layout is restated
from upstream and marked "unpinned" in the schema.  tests/test_proof_json.py checks the JSON the library emits against
this file key by key, in order."""
import json
import os
import re

REF = "/root/reference/crates"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def struct_fields(path, name):
    s = open(path).read()
    m = re.search(r"pub struct " + name + r"(?:<[^>]*>)? \{(.*?)\n\}", s, re.S)
    body = re.sub(r"//[^\n]*", "", m.group(1))
    return re.findall(r"pub (\w+):", body)


def main():
    p = f"{REF}/prover/src"
    mod = open(f"{p}/components/opcodes/mod.rs").read()
    macro = mod[mod.index("\ndefine_opcodes!(") + 1:]          # the invocation, not the macro_rules definition
    macro = re.sub(r"//[^\n]*", "", macro[:macro.index("\n);")])
    opcode_modules = re.findall(r"\],\s*(\w+)\s*\)", macro)
    assert len(opcode_modules) == 26
    vm = struct_fields(f"{REF}/common/src/state.rs", "State")     # `use cairo_m_common::State as VmRegisters` (adapter/mod.rs:10)
    schema = {
        "Proof": struct_fields(f"{p}/lib.rs", "Proof"),
        "Claim": struct_fields(f"{p}/components/mod.rs", "Claim"),
        "InteractionClaim": struct_fields(f"{p}/components/mod.rs", "InteractionClaim"),
        "opcodes": opcode_modules,
        "component_claim": struct_fields(f"{p}/components/memory.rs", "Claim"),
        "component_interaction_claim": struct_fields(f"{p}/components/memory.rs", "InteractionClaim"),
        "PublicData": struct_fields(f"{p}/public_data.rs", "PublicData"),
        "PublicEntries": struct_fields(f"{p}/public_data.rs", "PublicEntries"),
        "VmRegisters": vm,
        "stark_proof_unpinned": {
            "StarkProof": ["config", "commitments", "sampled_values", "decommitments", "queried_values", "proof_of_work", "fri_proof"],
            "PcsConfig": ["pow_bits", "fri_config"],
            "FriConfig": ["log_blowup_factor", "log_last_layer_degree_bound", "n_queries"],
            "MerkleDecommitment": ["hash_witness", "column_witness"],
            "FriProof": ["first_layer", "inner_layers", "last_layer_poly"],
            "FriLayerProof": ["fri_witness", "decommitment", "commitment"],
            "LinePoly": ["coeffs", "log_size"]},
    }
    out = os.path.join(ROOT, "tests", "golden", "proof_schema.json")
    json.dump(schema, open(out, "w"), indent=1)
    print(json.dumps(schema, indent=1)[:1500])
    print("wrote", out)


if __name__ == "__main__":
    main()
