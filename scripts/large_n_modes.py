#!/usr/bin/env python3
"""at the large configurations, three ctxs on three streams (sipp_instance_prove) against ONE ctx with the three proofs
back to back on one arena (sipp_instance_prove with three equal handles).  GPU box: large_n_modes.py [n ...]  (plain AIRs: the hardened
n = 4096 instance does not fit three arenas)"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sipp_amd  # noqa: E402

out = {}
for n in [int(x) for x in (sys.argv[1:] or ["1024", "4096"])]:
    d = np.load(os.path.join(ROOT, "tests", "golden", "sipp_n%d_ios.npz" % n))
    ios = [d["g1"], d["g2"], d["fq12"]]
    res = {}
    for hardened in (False, True):
        for single in (False, True):
            L = sipp_amd.lib()
            ws = [L.sipp_workspace_bytes(k + 4 if (hardened and k < 2) else k, ios[k].shape[0]) for k in range(3)]
            need = max(ws) if single else sum(ws)
            key = "%s_%s" % ("hardened" if hardened else "plain", "one_ctx" if single else "three_ctxs")
            if need > 0.93 * sipp_amd._lib.device_memory_bytes(0):
                res[key] = {"skipped": "arenas of %.0f GB do not fit the card" % (need / 2**30)}
                continue
            inst = sipp_amd.Instance([a.shape[0] for a in ios], hardened=hardened, single_ctx=single)
            try:
                inst.prove(ios)
                for c in inst.distinct_ctxs():
                    c.profile(True)
                    c.profile_reset()
                torch.cuda.synchronize()
                t = time.perf_counter()
                K = 3
                for _ in range(K):
                    inst.prove(ios)
                inst.sync()
                ms = 1e3 * (time.perf_counter() - t) / K
                prof = {}
                for c in inst.distinct_ctxs():
                    for k, v in c.profile_report().items():
                        prof[k] = prof.get(k, 0.0) + v["ms"] / K
                res[key] = {"ms_per_instance": round(ms, 1), "arena_GB": round(need / 2**30, 1),
                            "event_ms": {k: round(prof.get(k, 0.0), 1) for k in ("z_phase_a", "lookup_hist", "merkle_subtree", "poseidon_leaves")}}
            finally:
                inst.close()
    out["n=%d" % n] = res
    print(json.dumps({("n=%d" % n): res}), flush=True)
