#!/usr/bin/env python3
"""One-off stress run of the outer plonky2 prover with gates as data (sipp_plonk_prove_gates, GPU box): RANDOM circuit shapes of
tools/plonk_synth.py (wire and routed-wire counts, 2^10 .. 2^13 rows, witness seed, public inputs), random challenge counts and
FRI parameters (blowup 8 -- the degree-8 gate filters need it --, cap height, arity, final polynomial size, queries); every proof word
for word against oracle/plonk_gates.c and through its verifier.  usage: stress_plonk_gates.py [first_seed=1500] [count=16]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import plonk_synth as ps  # noqa: E402
import sipp_amd  # noqa: E402
from sipp_amd._lib import to_device  # noqa: E402
from tests import _oracle, _verify  # noqa: E402
from tests.test_gpu_fri_generic import to_params  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
count = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ctx = sipp_amd.Ctx(workspace_bytes=6 << 30)
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    log_n = int(rng.integers(10, 14))
    num_routed = int(rng.integers(8, 81))
    num_wires = int(max(num_routed, 33) + rng.integers(0, 60))
    challenges = int(rng.integers(1, 4))
    pis = [int(x) for x in rng.integers(0, 1 << 40, size=int(rng.integers(1, 9)))]
    fpb = int(rng.integers(0, 6))
    arity = int(rng.integers(1, 5))
    cap_h = int(rng.integers(0, 6))
    nq = int(rng.integers(1, 30))
    tag = "seed %d rows 2^%d wires %d routed %d challenges %d pis %d cap %d arity 2^%d final 2^%d queries %d" % (
        seed, log_n, num_wires, num_routed, challenges, len(pis), cap_h, arity, fpb, nq)
    circ = ps.circuit(num_wires, num_routed)
    pih = [int(x) for x in _oracle.hash_no_pad(np.array(pis, dtype=np.uint64))]
    wires, cs, gate = ps.witness(circ, log_n, seed, pih)
    op, gp = _oracle.plonk_params(num_routed, 8, challenges), sipp_amd.PlonkParams(num_routed, 8, challenges)
    ofp = _oracle.fri_params(rate_bits=3, cap_height=cap_h, pow_bits=int(rng.integers(0, 12)), num_queries=nq, pow_rule=int(rng.integers(0, 2)),
                             hiding=0, arity_bits=arity, final_poly_bits=fpb, degree_bits=log_n)
    digest = tuple(int(x) for x in rng.integers(0, 1 << 60, size=4))
    try:
        ref = _oracle.plonk_prove_gates(wires, cs, log_n, op, ofp, circ, digest, pis)
    except RuntimeError as e:
        ref, oerr = None, str(e)
    try:
        got = ctx.plonk_prove_gates(to_device(wires), to_device(cs), log_n, gp, to_params(ofp), sipp_amd.PlonkCircuit.from_dict(circ), digest, pis)
    except sipp_amd.SippError as e:
        if e.code == -7 and ref is not None:
            print("declined " + tag + " -- " + str(e)[:90], flush=True)
        else:
            print(("both refuse " if ref is None else "GPU-ONLY refusal ") + tag + " -- " + str(e)[:90], flush=True)
            bad += ref is not None
        continue
    if ref is None:
        bad += 1
        print("MISMATCH (oracle refuses: %s, GPU proves) %s" % (oerr, tag), flush=True)
        continue
    cs_cap = _oracle.Batch(cs, log_n, rate_bits=3, cap_height=cap_h).cap
    ok = len(got) == len(ref) and bool((got == ref).all()) and _oracle.plonk_verify_gates(got, cs_cap, op, ofp, circ, digest) == 0
    ok = ok and _verify.lib_plonk_verify(got, cs_cap, op, ofp, circ, digest) == 0          # the library's own verifier
    bad += not ok
    print(("ok   " if ok else "MISMATCH ") + tag + " (%.0f s)" % (time.time() - t0), flush=True)
ctx.close()
print("done: %d seeds, %d mismatches" % (count, bad))
sys.exit(1 if bad else 0)
