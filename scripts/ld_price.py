#!/usr/bin/env python3
"""Level L-D priced from the verifier's side: for world = 1, 2, 4, 8 ranks, the number of STARK proofs of ONE
instance, their total words, FRI query openings and Poseidon permutations a verifier performs (sipp_amd/proof_cost.py: counted from
the proofs' shapes, no GPU) -- what the recursive verifier behind reference src/verifier_circuit.rs:133-147 has to absorb.
usage: ld_price.py [n ...]   -> one JSON object per n"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sipp_amd import proof_cost  # noqa: E402

LISTS = {128: [127, 127, 14], 1024: [1023, 1023, 20], 4096: [4095, 4095, 24]}
for n in [int(a) for a in sys.argv[1:]] or [128, 1024, 4096]:
    rows = [proof_cost.instance_price(LISTS[n], w) for w in (1, 2, 4, 8)]
    base = rows[0]
    for r in rows:
        r["words_vs_world1"] = round(r["proof_words"] / base["proof_words"], 2)
        r["verifier_hashes_vs_world1"] = round(r["verifier_hashes"] / base["verifier_hashes"], 2)
    print(json.dumps({"n": n, "air": "hardened (kinds 4 / 5 / 2)", "stark_config": "rate_bits=1 cap=4 arity=16 queries=84", "by_world": rows}))
