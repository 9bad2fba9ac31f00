#!/usr/bin/env python3
"""One-off stress of the hardened G1 / G2 AIRs' exceptional rows (GPU box): random records of the two families that the plain AIR cannot
prove -- offset = [2^i - (e mod 2^i)] x (the accumulator meets the running power on add row i) and offset = -[2^i + (e mod 2^i)] x (it
meets MINUS the running power: the identity state bit) -- with random x, random 256-bit e with bit i set, random i; mixed with ordinary
records in one trace.  GPU trace against the oracle's cell for cell, the oracle's row check on every row of the exceptional records,
and one proof per kind and seed word for word.  usage: stress_hardened.py [first_seed=1] [count=6]"""
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sipp_amd  # noqa: E402
from sipp_amd._lib import to_host  # noqa: E402
from oracle.py import bn254 as bn  # noqa: E402
from oracle.py import sipp_native as sn  # noqa: E402
from tests import _oracle  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ctx = sipp_amd.Ctx(workspace_bytes=4 << 30)
bad = 0
for seed in range(first, first + count):
    rnd = random.Random(seed)
    t0 = time.time()
    for base in (0, 1):
        mul, neg, to_u32, gen = ((bn.g1_mul, bn.g1_neg, bn.g1_to_u32, bn.G1), (bn.g2_mul, bn.g2_neg, bn.g2_to_u32, bn.G2))[base]
        recs, what = [], []
        for r in range(4):
            x = mul(gen, rnd.randrange(1, bn.R))
            fam = rnd.randrange(3)
            i = rnd.choice([0, 1, 2, rnd.randrange(3, 255), 254])
            e = (rnd.randrange(1 << 256) | (1 << i)) if rnd.randrange(2) else ((1 << i) | rnd.randrange(1 << i) if i else 1)
            low = e % (1 << i)
            if fam == 0:      # ordinary
                o = mul(gen, rnd.randrange(1, bn.R))
                k = None
                out = bn.g1_add(o, mul(x, e % bn.R)) if base == 0 else bn.g2_add(o, mul(x, e % bn.R))
            elif fam == 1:    # R = P on add row i
                k = (1 << i) - low
                o, out = mul(x, k), mul(x, (k + e) % bn.R)
            else:             # R = -P on add row i
                k = (1 << i) + low
                o, out = neg(mul(x, k)), mul(x, (e - k) % bn.R)
            if out is None:   # an output at the identity has no record
                continue
            recs.append(to_u32(x) + to_u32(o) + sn.exp_to_u32(e) + to_u32(out))
            what.append((fam, i))
        if not recs:
            continue
        recs = np.array(recs, dtype=np.uint32)
        kind = base + 4
        try:
            ref = _oracle.Trace(kind, recs)
        except RuntimeError:
            # what stays unprovable: the accumulator meets the running power on the LAST add row (all exponent bits
            # from i up are set) -- both sides have to refuse
            try:
                ctx.prove(kind, recs)
                bad += 1
                print("seed %d kind %d: the oracle refuses, the GPU proves, records %s" % (seed, kind, what))
            except sipp_amd.SippError as ex:
                assert ex.code == -8
                print("seed %d kind %d: refused by both (%s)" % (seed, kind, what))
            continue
        want = ref.array()
        got = to_host(ctx.trace_build(kind, recs))
        if not (got == want).all():
            bad += 1
            print("seed %d kind %d: %d cells differ %s, records %s" % (seed, kind, int((got != want).sum()), np.argwhere(got != want)[:4].tolist(), what))
            continue
        rows = [ref.check_row(r) for r in range(512 * len(recs))]
        if any(x != -1 for x in rows):
            bad += 1
            print("seed %d kind %d: a row of the oracle's own trace breaks constraint %s, records %s" % (seed, kind, [x for x in rows if x != -1][:3], what))
            continue
        pf = ctx.prove(kind, recs)
        if not ((pf == _oracle.stark_prove(kind, recs)).all() and _oracle.stark_verify(pf) == 0):
            bad += 1
            print("seed %d kind %d: proof differs or is refused, records %s" % (seed, kind, what))
        if any(f for f, _ in what):
            try:
                ctx.prove(base, recs)
                bad += 1
                print("seed %d kind %d: the plain AIR proved an exceptional record %s" % (seed, base, what))
            except sipp_amd.SippError as ex:
                assert ex.code == -8
    print("seed %d ok (%.0f s)" % (seed, time.time() - t0), flush=True)
print("done: %d seeds, %d failures" % (count, bad))
sys.exit(1 if bad else 0)
