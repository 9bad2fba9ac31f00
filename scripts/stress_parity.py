#!/usr/bin/env python3
"""One-off stress run of the proof parity (GPU box): fresh random obligations per seed -- random points, random Fq12 elements,
random exponents incl. edge patterns, random record counts 1..9 -- outputs from sipp_exp_outputs, then every proof word for
word against the CPU oracle, through its verifier and through the library's own; every fourth seed also 1 .. 3 final-pairing records (kind 6).  usage: stress_parity.py [first_seed=100] [count=30]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sipp_amd  # noqa: E402
from oracle.py import bn254 as bn  # noqa: E402
from tests import _oracle  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 30
ctx = sipp_amd.Ctx(workspace_bytes=8 << 30)
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)

    def scalar():
        mode = rng.integers(0, 8)
        if mode == 0:
            return int(rng.integers(1, 4))                      # tiny
        if mode == 1:
            return bn.R - int(rng.integers(1, 4))               # just under r
        if mode == 2:
            return 1 << int(rng.integers(1, 254))               # one bit
        if mode == 3:
            return (1 << 256) - 1 - int(rng.integers(0, 1 << 20))  # 256-bit exponents are legal (U256Target)
        if mode == 4:
            return [0, 1, bn.R, bn.R + 1, (1 << 256) - 1, 1 << 255][int(rng.integers(0, 6))]
        return int.from_bytes(rng.bytes(32), "little") % bn.R or 1

    def fq():
        return int.from_bytes(rng.bytes(32), "little") % bn.P

    def rand_f12():
        mode = rng.integers(0, 8)
        if mode == 0:
            return [1] + [0] * 11                               # one
        if mode == 1:
            return [0] * 12                                     # zero
        if mode == 2:
            return [bn.P - 1] + [0] * 11                        # minus one
        if mode == 3:
            return [fq()] + [0] * 11                            # a base-field element
        if mode == 4:
            return [0] * 6 + [1] + [0] * 5                      # w^6
        return [fq() for _ in range(12)]

    def words(e):
        return [(e >> (32 * i)) & 0xFFFFFFFF for i in range(8)]

    n1, n2, n12 = (int(rng.integers(1, 10)) for _ in range(3))
    g1 = [bn.g1_to_u32(bn.g1_mul(bn.G1, scalar() % bn.R or 1)) + bn.g1_to_u32(bn.g1_mul(bn.G1, scalar() % bn.R or 2)) + words(scalar() % (1 << 256)) + [0] * 16
          for _ in range(n1)]
    g2 = [bn.g2_to_u32(bn.g2_mul(bn.G2, scalar() % bn.R or 1)) + bn.g2_to_u32(bn.g2_mul(bn.G2, scalar() % bn.R or 2)) + words(scalar() % (1 << 256)) + [0] * 32
          for _ in range(n2)]
    f12 = [bn.f12_to_u32(rand_f12()) + bn.f12_to_u32(rand_f12()) + words(scalar()) + [0] * 96 for _ in range(n12)]
    if rng.integers(0, 4) == 0:      # a point off its curve in one record: both sides must refuse
        k = int(rng.integers(0, n1))
        g1[k] = g1[k][:8] + words(fq()) + g1[k][16:]
    if rng.integers(0, 4) == 0:
        k = int(rng.integers(0, n2))
        g2[k] = g2[k][:32 + 16] + words(fq()) + g2[k][32 + 24:]
    for kind, recs in ((0, g1), (1, g2), (2, f12), (4, g1), (5, g2)):      # 4 / 5: the hardened AIRs on the same records
        try:
            ios = ctx.exp_outputs(kind, np.array(recs, dtype=np.uint32))
        except sipp_amd.SippError as e:      # an exceptional addition (negligible for random inputs): report, do not stop
            print("seed %d kind %d: exp_outputs refused: %s" % (seed, kind, e))
            continue
        try:
            ref = _oracle.stark_prove(kind, ios)
        except RuntimeError as e:
            # an exceptional addition (accumulator = +- running power; likely with the tiny scalars above): no witness exists,
            # the GPU prover must refuse the same records
            try:
                ctx.prove(kind, ios)
                bad += 1
                print("MISMATCH seed %d kind %d: oracle refuses (%s), GPU proves" % (seed, kind, e))
            except sipp_amd.SippError as ge:
                print("seed %d kind %d: both refuse (%s | %s)" % (seed, kind, e, str(ge)[:80]))
            continue
        try:
            got = ctx.prove(kind, ios)
        except sipp_amd.SippError as ge:
            bad += 1
            print("MISMATCH seed %d kind %d: oracle proves, GPU refuses (%s)" % (seed, kind, str(ge)[:100]))
            np.save(os.path.join(ROOT, "gpurun_out", "stress_fail_seed%d_kind%d.npy" % (seed, kind)), ios)
            continue
        ok = len(got) == len(ref) and bool((got == ref).all()) and _oracle.stark_verify(got) == 0 and sipp_amd.stark_verify(got) == 0
        if not ok:
            bad += 1
            print("MISMATCH seed %d kind %d (%d records)" % (seed, kind, len(recs)))
    if seed % 4 == 0:                # the final-pairing AIR (kind 6): 1 .. 3 records of random / edge multiples of the generators
        npair = int(rng.integers(1, 4))
        prec = np.zeros((npair, 144), dtype=np.uint32)
        for k in range(npair):
            prec[k, :48] = bn.g1_to_u32(bn.g1_mul(bn.G1, scalar() % bn.R or 1)) + bn.g2_to_u32(bn.g2_mul(bn.G2, scalar() % bn.R or 5))
        pios = ctx.exp_outputs(6, prec)
        got = ctx.prove(6, pios)
        ref = _oracle.stark_prove(6, pios)
        if not (len(got) == len(ref) and bool((got == ref).all()) and _oracle.stark_verify(got) == 0 and sipp_amd.stark_verify(got) == 0):
            bad += 1
            print("MISMATCH seed %d kind 6 (%d records)" % (seed, npair))
    print("seed %d ok (%d/%d/%d records, %.0f s)" % (seed, n1, n2, n12, time.time() - t0), flush=True)
print("done: %d seeds, %d mismatches" % (count, bad))
sys.exit(1 if bad else 0)
