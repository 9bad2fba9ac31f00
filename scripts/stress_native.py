#!/usr/bin/env python3
"""One-off stress run of the native chain (GPU box): sipp_prove_native / sipp_verify_native against oracle/py/sipp_native.py on
small instances whose points are RELATED (small multiples of the generators, repeated points, A_i = A_j, B_i = -B_j) -- the
inputs where incomplete group formulas would show.  usage: stress_native.py [first_seed=500] [count=8]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sipp_amd  # noqa: E402
from oracle.py import bn254 as bn  # noqa: E402
from oracle.py import sipp_native as sn  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 500
count = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ctx = sipp_amd.Ctx(workspace_bytes=4 << 30)
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    n = int([2, 4, 4, 8][int(rng.integers(0, 4))])

    def sc():
        mode = rng.integers(0, 4)
        if mode == 0:
            return int(rng.integers(1, 6))
        if mode == 1:
            return bn.R - int(rng.integers(1, 6))
        return int.from_bytes(rng.bytes(32), "little") % bn.R or 1

    a = [sc() for _ in range(n)]
    b = [sc() for _ in range(n)]
    if rng.integers(0, 2):
        a[-1] = a[0]                        # a repeated point
    if rng.integers(0, 2):
        b[-1] = bn.R - b[0]                 # B_last = -B_0
    A = [bn.g1_mul(bn.G1, k) for k in a]
    B = [bn.g2_mul(bn.G2, k) for k in b]
    An = np.array([bn.g1_to_u32(p) for p in A], dtype=np.uint32)
    Bn = np.array([bn.g2_to_u32(p) for p in B], dtype=np.uint32)
    want_proof = sn.sipp_prove_native(A, B)
    want = np.array([w for m in want_proof for w in bn.f12_to_u32(m)], dtype=np.uint32)
    try:
        got = ctx.prove_native(An, Bn)
    except sipp_amd.SippError as e:
        bad += 1
        print("MISMATCH seed %d (n = %d): GPU refuses: %s" % (seed, n, str(e)[:120]))
        continue
    ok = got.size == want.size and bool((got.ravel() == want).all())
    if ok:
        accepted, st, ios = ctx.verify_native(An, Bn, got)
        ok_ref, st_ref, obl = sn.sipp_verify_native(A, B, want_proof)
        want_ios = sn.io_records(obl)
        ok = (bool(accepted) == bool(ok_ref) and bool((np.asarray(st).ravel() == np.array(sn.statement_to_u32(st_ref), dtype=np.uint32)).all())
              and all(g.shape == w.shape and bool((g == w).all()) for g, w in zip(ios, want_ios)))
    if not ok:
        bad += 1
        print("MISMATCH seed %d (n = %d, a = %s, b = %s)" % (seed, n, [k if k < 10 else "..." for k in a], [k if k < 10 else "..." for k in b]))
    print("seed %d ok=%s (n = %d, %.0f s)" % (seed, ok, n, time.time() - t0), flush=True)
print("done: %d seeds, %d mismatches" % (count, bad))
sys.exit(1 if bad else 0)
