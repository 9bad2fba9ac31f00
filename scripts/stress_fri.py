#!/usr/bin/env python3
"""One-off stress run of the generic layer (GPU box): random FriParams (blowup 2 / 4 / 8, cap heights, constant or mixed arities
2 .. 16, final polynomial sizes, both proof-of-work rules, query counts), random numbers of oracles with random widths (1 .. 4
columns = unhashed leaves included), random salting, random batches of column ranges -- sipp_commit_batch_ex and
sipp_fri_prove_openings against oracle/fri.c word for word, then the oracle's verifier.
usage: stress_fri.py [first_seed=700] [count=40]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sipp_amd  # noqa: E402
from sipp_amd._lib import to_device  # noqa: E402
from tests import _oracle, _verify  # noqa: E402
from tests.test_gpu_fri_generic import gpu_challenger, to_params  # noqa: E402

P = _oracle.P
first = int(sys.argv[1]) if len(sys.argv) > 1 else 700
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ctx = sipp_amd.Ctx(workspace_bytes=4 << 30)
bad = skipped = declined = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    log_n = int(rng.integers(10, 15))       # the GPU layer supports degree bits 10 .. 24
    rate_bits = int(rng.integers(1, 4))
    n, m = 1 << log_n, 1 << (log_n + rate_bits)
    n_or = int(rng.integers(1, 5))
    widths = [int([1, 2, 4, 5, 8, 9, 17, 33][int(rng.integers(0, 8))]) for _ in range(n_or)]
    salted = [bool(rng.integers(0, 2)) for _ in range(n_or)]
    mixed = bool(rng.integers(0, 3) == 0)
    pow_rule = int(rng.integers(0, 2))
    nq = int(rng.integers(1, 13))
    pow_bits = int(rng.integers(0, 11))
    if mixed:
        arities, left = [], log_n
        while left > 0 and len(arities) < 6 and rng.integers(0, 4):
            a = int(rng.integers(1, min(4, left) + 1))
            arities.append(a)
            left -= a
        if not arities:
            arities = [min(2, log_n)]
        final_bits = log_n - sum(arities)
        cap_height = int(rng.integers(0, min(5, final_bits + rate_bits) + 1))
        fp = _oracle.fri_params(rate_bits=rate_bits, cap_height=cap_height, pow_bits=pow_bits, num_queries=nq, pow_rule=pow_rule, hiding=1,
                                arities=arities)
        desc = "arities %s" % arities
    else:
        arity = int(rng.integers(1, 5))
        final_poly_bits = int(rng.integers(0, 6))
        cap_height = int(rng.integers(0, 6))
        fp = _oracle.fri_params(rate_bits=rate_bits, cap_height=cap_height, pow_bits=pow_bits, num_queries=nq, pow_rule=pow_rule, hiding=1,
                                arity_bits=arity, final_poly_bits=final_poly_bits, degree_bits=log_n)
        desc = "arity %d final %d rounds %d" % (arity, final_poly_bits, fp.n_rounds)
    tag = "seed %d: log_n %d blowup %d cap %d %s pow %d/%d q %d widths %s salted %s" % (
        seed, log_n, 1 << rate_bits, cap_height, desc, pow_bits, pow_rule, nq, widths, [int(x) for x in salted])
    try:
        oracles = []
        for k in range(n_or):
            vals = _oracle.rand_field(rng, (widths[k], n))
            salt = _oracle.rand_field(rng, (4, m)) if salted[k] else None
            oracles.append(_oracle.SaltedBatch(vals, log_n, rate_bits, cap_height, from_values=(k % 2 == 0), salt=salt))
        zeta = tuple(int(x) for x in _oracle.rand_field(rng, 2))
        w = pow(1753635133440165772, 1 << (32 - log_n), P)
        gz = (zeta[0] * w % P, zeta[1] * w % P)
        batches = [(zeta, [(k, 0, widths[k]) for k in range(n_or)])]
        sub = []
        for k in range(n_or):
            if rng.integers(0, 2):
                lo = int(rng.integers(0, widths[k]))
                hi = int(rng.integers(lo + 1, widths[k] + 1))
                sub.append((k, lo, hi))
        if sub:
            batches.append((gz, sub))
        gch, och = gpu_challenger([seed, 1, 2])
        ref = _oracle.fri_prove_openings(oracles, batches, log_n, fp, och)
    except Exception as e:      # noqa: BLE001 -- a configuration the oracle itself refuses (e.g. cap higher than the last layer)
        skipped += 1
        print(tag + " -- oracle refuses: %s" % str(e)[:80], flush=True)
        continue
    try:
        devs, keep = [], []
        for k, o in enumerate(oracles):
            coeffs = o.coeffs
            if k % 2 == 0:
                vals = coeffs.copy()
                L = _oracle.load()
                for c in range(vals.shape[0]):
                    L.orc_fft(vals[c], log_n)
                data, from_coeffs = vals, False
            else:
                data, from_coeffs = coeffs, True
            salt = None if o.salt is None else to_device(o.salt)
            od, cap, bufs = ctx.commit_ex(to_device(data), log_n, rate_bits, cap_height, from_coeffs=from_coeffs, salt=salt)
            assert (cap == o.cap).all(), "cap of oracle %d" % k
            devs.append(od)
            keep.append(bufs)
        got = ctx.fri_prove_openings(devs, batches, log_n, to_params(fp), gch)
        ok = len(got) == len(ref) and bool((got == ref).all())
        ok = ok and [gch.state[i] for i in range(12)] == [och.state[i] for i in range(12)]
        ok = ok and _oracle.fri_verify_openings(got, [o.cap for o in oracles], [o.ncols for o in oracles], [o.n_salt for o in oracles],
                                                batches, log_n, fp, _oracle.challenger([seed, 1, 2])) == 0
        # ... and the library's own verifier (sipp_fri_verify_openings), transcript included
        lib_stage, lib_ch = _verify.lib_fri_verify(got, [o.cap for o in oracles], [o.ncols for o in oracles], [o.n_salt for o in oracles], batches, log_n, fp,
                                                   _oracle.challenger([seed, 1, 2]))
        ok = ok and lib_stage == 0 and lib_ch == bytes(och)
    except sipp_amd.SippError as e:
        if e.code == -7:        # SIPP_E_UNSUPPORTED: outside the documented range of the GPU layer (e.g. a layer of < 16 values)
            declined += 1
            print("declined " + tag + " -- %s" % str(e)[:90], flush=True)
            continue
        ok = False
        tag += " -- GPU: %s" % str(e)[:100]
    except AssertionError as e:
        ok = False
        tag += " -- GPU: %s" % str(e)[:100]
    if not ok:
        bad += 1
    print(("ok   " if ok else "MISMATCH ") + tag + " (%.0f s)" % (time.time() - t0), flush=True)
print("done: %d seeds, %d mismatches, %d refused by the oracle, %d declined by the GPU layer (unsupported range)" % (count, bad, skipped, declined))
sys.exit(1 if bad else 0)
