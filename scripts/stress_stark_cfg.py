#!/usr/bin/env python3
"""One-off stress run of the three STARK provers under RANDOM sipp_stark_config values (GPU box): blowup 2 / 4 / 8, cap height,
proof-of-work bits and rule, Fiat-Shamir start and lookup-challenge rule (round 5: fs_rule, lookup_rule), reduction arity 2 .. 16, final
polynomial size, query count -- plain and hardened curve kinds, on a random number of records of the n = 8 fixture, every proof word
for word against the CPU oracle and through its verifier; a configuration one side refuses must be refused by the other.  usage: stress_stark_cfg.py [first_seed=900] [count=30] [max_records: 9 .. max records of the n = 1024 fixture instead (2^13 .. rows)]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sipp_amd  # noqa: E402
from tests import _oracle  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 900
count = int(sys.argv[2]) if len(sys.argv) > 2 else 30
maxrec = int(sys.argv[3]) if len(sys.argv) > 3 else 0
d = np.load(os.path.join(ROOT, "tests", "golden", "sipp_n1024_ios.npz" if maxrec else "sipp_n8_ios.npz"))
lists = [d["g1"][:maxrec], d["g2"][:maxrec], d["fq12"][:maxrec]] if maxrec else [d["g1"], d["g2"], d["fq12"]]
L = sipp_amd.lib()
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    cfg, ocfg = sipp_amd.default_config(), _oracle.default_config()
    vals = dict(rate_bits=int(rng.integers(1, 4)), cap_height=int(rng.integers(0, 7)), pow_bits=int(rng.integers(0, 15)),
                arity_bits=int(rng.integers(1, 5)), final_poly_bits=int(rng.integers(0, 8)), num_queries=int(rng.integers(1, 40)),
                pow_rule=int(rng.integers(0, 2)), fs_rule=int(rng.integers(0, 2)), lookup_rule=int(rng.integers(0, 2)))
    for c in (cfg, ocfg):
        for k, v in vals.items():
            setattr(c, k, v)
    kind = int(rng.integers(0, 5))
    kind = kind if kind < 3 else kind + 1                      # 0 G1, 1 G2, 2 Fq12, 4 / 5 the hardened G1 / G2 AIRs
    num = int(rng.integers(9 if maxrec else 1, lists[kind & 3].shape[0] + 1))
    ios = np.ascontiguousarray(lists[kind & 3][:num])
    tag = "seed %d kind %d records %d %s" % (seed, kind, num, vals)
    try:
        ref = _oracle.stark_prove(kind, ios, ocfg)
    except RuntimeError as e:
        ref = None
        oerr = str(e)
    try:
        ws = L.sipp_workspace_bytes_cfg(kind, num, C.byref(cfg))
        ctx = sipp_amd.Ctx(cfg=cfg, workspace_bytes=max(ws, 1 << 20))
    except sipp_amd.SippError as e:
        print(("both refuse " if ref is None else "GPU-ONLY refusal (ctx) ") + tag + " -- " + str(e)[:80], flush=True)
        bad += ref is not None
        continue
    try:
        got = ctx.prove(kind, ios)
        if ref is None:
            bad += 1
            print("MISMATCH (oracle refuses: %s, GPU proves) %s" % (oerr, tag), flush=True)
        else:
            ok = len(got) == len(ref) and bool((got == ref).all()) and _oracle.stark_verify(got, ocfg) == 0
            bad += not ok
            print(("ok   " if ok else "MISMATCH ") + tag + " (%.0f s)" % (time.time() - t0), flush=True)
    except sipp_amd.SippError as e:
        if e.code == -7:          # SIPP_E_UNSUPPORTED: outside the GPU layer's documented range (FRI layers of < 16 values)
            print("declined " + tag + " -- " + str(e)[:90], flush=True)
        else:
            print(("both refuse " if ref is None else "GPU-ONLY refusal ") + tag + " -- " + str(e)[:80], flush=True)
            bad += ref is not None
    finally:
        ctx.close()
print("done: %d seeds, %d mismatches" % (count, bad))
sys.exit(1 if bad else 0)
