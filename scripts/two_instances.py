"""Experiment: is one n=128 instance (3 streams) saturating the GPU?  Run 1 vs 2 instances concurrently."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sipp_amd
from concurrent.futures import ThreadPoolExecutor
d = np.load("tests/golden/sipp_n128_ios.npz"); ios = [d["g1"], d["g2"], d["fq12"]]
L = sipp_amd.lib()
for ninst in (1, 2, 3):
    ctxs = []
    for i in range(ninst):
        for k, pr in enumerate(("low", "high", "high")):
            if pr: os.environ["SIPP_STREAM_PRIORITY"] = pr
            else: os.environ.pop("SIPP_STREAM_PRIORITY", None)
            ctxs.append((k, sipp_amd.Ctx(workspace_bytes=L.sipp_workspace_bytes(k, ios[k].shape[0]))))
    pool = ThreadPoolExecutor(max_workers=3 * ninst)
    run = lambda: list(pool.map(lambda kc: kc[1].prove(kc[0], ios[kc[0]]), sorted(ctxs, key=lambda kc: -(kc[0] == 1))))
    run()
    t = time.perf_counter(); K = 5
    for _ in range(K): run()
    dt = (time.perf_counter() - t) / K
    print("instances %d: %.1f ms per step, %.1f ms per instance, %.0f pairings/s" % (ninst, dt * 1e3, dt * 1e3 / ninst, 128 * ninst / dt))
    for _, c in ctxs: c.close()
