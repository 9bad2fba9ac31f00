"""Price of the hardened G1 / G2 AIRs (kinds 4 / 5) at the n = 128 size: each proof alone on the GPU, plain against hardened."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, sipp_amd
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "sipp_n128_ios.npz"))
L = sipp_amd.lib()
for base, key in ((0, "g1"), (1, "g2")):
    ios = d[key]
    ctx = sipp_amd.Ctx(workspace_bytes=L.sipp_workspace_bytes(base + 4, ios.shape[0]))
    for kind in (base, base + 4):
        ctx.prove(kind, ios)
        t = time.perf_counter()
        for _ in range(5):
            pf = ctx.prove(kind, ios)
        ms = (time.perf_counter() - t) / 5 * 1e3
        print("kind %d %s shape %s: %.2f ms, %d words" % (kind, key, ctx.shape(kind, ios.shape[0]), ms, len(pf)), flush=True)
    ctx.close()
