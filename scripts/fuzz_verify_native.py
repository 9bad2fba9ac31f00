#!/usr/bin/env python3
"""One-off fuzz of sipp_verify_native (GPU box): valid inputs with DAMAGED proofs (a message replaced by zero / one / a random Fq12
element, every message random, messages swapped) -- the accept bit, the SIPPStatement limbs and the three obligation lists must
equal the CPU restatement's (oracle/py/sipp_native.py) in every case.  40 cases, no mismatch (round 2)."""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import sipp_amd
from oracle.py import bn254 as bn, sipp_native as sn
ctx = sipp_amd.Ctx(workspace_bytes=2 << 30)
bad = 0
for seed in range(40):
    rng = np.random.default_rng(seed)
    n = [2, 4, 8][seed % 3]
    A = [bn.g1_mul(bn.G1, int(rng.integers(1, 1 << 40))) for _ in range(n)]
    B = [bn.g2_mul(bn.G2, int(rng.integers(1, 1 << 40))) for _ in range(n)]
    good = sn.sipp_prove_native(A, B)
    proof = list(good)
    mode = seed % 4
    def rf():
        m = rng.integers(0, 4)
        if m == 0: return [0] * 12
        if m == 1: return [1] + [0] * 11
        return [int.from_bytes(rng.bytes(32), "little") % bn.P for _ in range(12)]
    if mode == 0: proof[int(rng.integers(0, len(proof)))] = rf()
    elif mode == 1: proof = [rf() for _ in proof]
    elif mode == 2: proof[0], proof[-1] = proof[-1], proof[0]
    An = np.array([bn.g1_to_u32(p) for p in A], dtype=np.uint32); Bn = np.array([bn.g2_to_u32(p) for p in B], dtype=np.uint32)
    pw = np.array([w for m in proof for w in bn.f12_to_u32(m)], dtype=np.uint32)
    ok_ref, st_ref, obl = sn.sipp_verify_native(A, B, proof)
    try:
        acc, st, ios = ctx.verify_native(An, Bn, pw)
    except sipp_amd.SippError as e:
        print("seed", seed, "GPU error", str(e)[:100]); bad += 1; continue
    want_ios = sn.io_records(obl)
    same = (bool(acc) == bool(ok_ref) and bool((np.asarray(st).ravel() == np.array(sn.statement_to_u32(st_ref), dtype=np.uint32)).all())
            and all(g.shape == w.shape and bool((g == w).all()) for g, w in zip(ios, want_ios)))
    if not same: bad += 1; print("MISMATCH seed", seed, "mode", mode, "acc", acc, ok_ref)
print("done, mismatches:", bad)
