#!/usr/bin/env python3
"""One-off stress run of the outer prover's witness generation and CircuitData (GPU box): RANDOM sizes (2^10 .. 2^15 rows), witness seeds,
chain lengths (0 = row-local generators; 1 .. 300 links: the level schedule), public inputs; the device's wire table cell for cell
against oracle/plonk_witness.c (itself = the numpy generator), graph replay and one-by-one launches alike; for the smaller sizes the
whole `data.prove(pw)` through sipp_circuit_build / _prove word for word against oracle/plonk_gates.c, and through both verifiers.
usage: stress_witness.py [first_seed=2600] [count=20]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import plonk_synth as ps  # noqa: E402
import sipp_amd  # noqa: E402
from sipp_amd._lib import to_device, to_host  # noqa: E402
from tests import _oracle  # noqa: E402
from tests.test_gpu_fri_generic import to_params  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 2600
count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
circ = ps.circuit(136, 80)
K, gens = circ["num_constants"], ps.generators(circ)
gc = sipp_amd.PlonkCircuit.from_dict(circ)
ctx = sipp_amd.Ctx(workspace_bytes=8 << 30)
L = sipp_amd.lib()
bad, proofs = 0, 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    log_n = int(rng.integers(10, 16))
    chain = int(rng.choice([0, 0, 1, 2, 5, 16, 64, 100, 300]))
    pis = [int(x) for x in rng.integers(0, 1 << 50, size=int(rng.integers(1, 12)))]
    pih = [int(x) for x in _oracle.hash_no_pad(np.array(pis, dtype=np.uint64))]
    tag = "seed %d rows 2^%d chain %d pis %d" % (seed, log_n, chain, len(pis))
    wires, cs, gate = ps.witness(circ, log_n, seed, pih, chain_len=chain)
    sc = ps.chain_schedule(log_n, chain) if chain else None
    blank = ps.blank_generated(circ, wires, gate, value=int(rng.integers(0, 1 << 62)), sched=sc)
    ref = _oracle.plonk_generate_witness_levels(blank, cs[:K], log_n, gens, pih, sc) if sc else _oracle.plonk_generate_witness(blank, cs[:K], log_n, gens, pih)
    ok = bool((ref == wires).all())
    d_cs = to_device(cs)
    sched = sipp_amd.PlonkSchedule.from_dict(sc) if sc else None
    for route in ((4, 0, 0) if sc else (0,)):
        L.sipp_ctx_set_kernel_routes(ctx.h, route)
        d_w = to_device(blank)
        if sc:
            ctx.plonk_generate_witness_levels(d_w, d_cs[:K], log_n, gens, pih, sched)
        else:
            ctx.plonk_generate_witness(d_w, d_cs[:K], log_n, gens, pih)
            ctx.sync()
        ok = ok and bool((to_host(d_w) == ref).all())
    L.sipp_ctx_set_kernel_routes(ctx.h, 0)
    note = ""
    if log_n <= 12:
        ofp = _oracle.fri_params(rate_bits=3, cap_height=int(rng.integers(0, 5)), pow_bits=int(rng.integers(0, 10)), num_queries=int(rng.integers(2, 20)),
                                 pow_rule=0, hiding=0, arity_bits=int(rng.integers(1, 5)), final_poly_bits=int(rng.integers(2, 6)), degree_bits=log_n)
        op, gp = _oracle.plonk_params(80, 8, 2), sipp_amd.PlonkParams(80, 8, 2)
        data = sipp_amd.CircuitData(ctx, log_n, gp, to_params(ofp), gc, cs, gens, sched=sc)
        try:
            pf = data.prove(blank, pis)
            digest = [int(x) for x in data.digest]
            want = _oracle.plonk_prove_gates(wires, cs, log_n, op, ofp, circ, digest, pis)
            same = len(pf) == len(want) and bool((pf == want).all())
            ver = data.verify(pf) == (0, 0) and _oracle.plonk_verify_gates(pf, data.cap, op, ofp, circ, digest) == 0
            ok = ok and same and ver
            proofs += 1
            note = "  proof %d words %s" % (len(pf), "= oracle, verified" if same and ver else "MISMATCH")
        finally:
            data.close()
    bad += not ok
    print(("ok   " if ok else "FAIL ") + tag + ("  levels %d copies %d" % (sc["n_levels"], len(sc["copy_src"])) if sc else "  row-local") + note, flush=True)
ctx.close()
print("stress_witness: %d cases (%d with whole proofs), %d mismatches, %.0f s" % (count, proofs, bad, time.time() - t0))
sys.exit(1 if bad else 0)
