#!/usr/bin/env python3
"""One-off fuzz of the flat proof encoding against the CPU oracle's verifier (no GPU): for the kinds added in round 3 (MapToG2 = 3,
hardened G1 / G2 = 4 / 5; any kind by argument) every header word and a random sample of the words of every section is changed --
+1, one random bit, and the second encoding w + p where it fits a u64 -- and the verifier has to refuse each changed proof: no word
of a proof is free, and no value has two encodings.  usage: fuzz_proof_words.py [kinds=3,4,5] [samples_per_section=40] [seed=1]"""
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "py"))
import bn254 as bn  # noqa: E402
from tests import _oracle  # noqa: E402

kinds = [int(k) for k in (sys.argv[1] if len(sys.argv) > 1 else "3,4,5").split(",")]
samples = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rnd = random.Random(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
P = _oracle.P


def words(e):
    return [(e >> (32 * i)) & 0xFFFFFFFF for i in range(8)]


def records(kind):
    if kind == 3:
        us = [(rnd.randrange(bn.P), rnd.randrange(bn.P)) for _ in range(3)]
        return _oracle.map_to_g2(np.array([bn.fq_to_u32(u[0]) + bn.fq_to_u32(u[1]) for u in us], dtype=np.uint32))
    base = kind & 3 if kind >= 4 else kind
    out = []
    for _ in range(2):
        e = rnd.randrange(1 << 256)
        if base == 0:
            x, o = bn.g1_mul(bn.G1, rnd.randrange(1, bn.R)), bn.g1_mul(bn.G1, rnd.randrange(1, bn.R))
            out.append(bn.g1_to_u32(x) + bn.g1_to_u32(o) + words(e) + bn.g1_to_u32(bn.g1_add(o, bn.g1_mul(x, e % bn.R))))
        elif base == 1:
            x, o = bn.g2_mul(bn.G2, rnd.randrange(1, bn.R)), bn.g2_mul(bn.G2, rnd.randrange(1, bn.R))
            out.append(bn.g2_to_u32(x) + bn.g2_to_u32(o) + words(e) + bn.g2_to_u32(bn.g2_add(o, bn.g2_mul(x, e % bn.R))))
        else:
            x, o = [rnd.randrange(bn.P) for _ in range(12)], [rnd.randrange(bn.P) for _ in range(12)]
            out.append(bn.f12_to_u32(x) + bn.f12_to_u32(o) + words(e) + bn.f12_to_u32(bn.f12_mul(o, bn.f12_pow(x, e))))
    return np.array(out, dtype=np.uint32)


total = accepted = 0
for kind in kinds:
    t0 = time.time()
    pf = _oracle.stark_prove(kind, records(kind))
    assert _oracle.stark_verify(pf) == 0
    W, Pc, Q = int(pf[4]), int(pf[5]), int(pf[6])
    caps = 16 + 3 * 64
    fri = caps + 2 * (2 * W + 2 * Pc + Q)
    rounds, flen = int(pf[8]), int(pf[9])
    pow_at = fri + rounds * 64 + 2 * flen
    n_pi = int(pf[3]) * {0: 48, 1: 96, 2: 296, 3: 48}[kind if kind < 4 else kind - 4]
    sections = {"header": (0, 16), "caps": (16, caps), "openings": (caps, fri), "fri caps + final polynomial": (fri, pow_at),
                "pow witness": (pow_at, pow_at + 1), "queries": (pow_at + 1, len(pf) - n_pi), "public inputs": (len(pf) - n_pi, len(pf))}
    for name, (a, b) in sections.items():
        idx = list(range(a, b)) if b - a <= max(16, samples) else sorted(rnd.sample(range(a, b), samples))
        bad = []
        for i in idx:
            w = int(pf[i])
            cands = {(w + 1) & (2**64 - 1), w ^ (1 << rnd.randrange(64))}
            if w + P < 2**64:
                cands.add(w + P)            # the same field element, second encoding
            for c in cands:
                q = pf.copy()
                q[i] = c
                total += 1
                if _oracle.stark_verify(q) == 0:
                    accepted += 1
                    bad.append((i, w, c))
        print("kind %d %-28s words [%d, %d): %d positions, %d accepted %s" % (kind, name, a, b, len(idx), len(bad), bad[:6]), flush=True)
    print("kind %d done in %.0f s" % (kind, time.time() - t0), flush=True)
print("changed proofs: %d, accepted: %d" % (total, accepted))
sys.exit(1 if accepted else 0)
