"""Per-kernel times of the trace fill of the three AIRs (serial, HIP-event bracketed): python scripts/perf_trace.py [n]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sipp_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = np.load(os.path.join(root, "tests/golden/sipp_n%d_ios.npz" % n))
L = sipp_amd.lib()
for kind, key in enumerate(("g1", "g2", "fq12")):
    ios = d[key]
    ctx = sipp_amd.Ctx(workspace_bytes=L.sipp_workspace_bytes(kind, ios.shape[0]))
    ctx.trace_build(kind, ios)
    ctx.profile(True)
    ctx.profile_reset()
    reps = 3
    for _ in range(reps):
        ctx.trace_build(kind, ios)
    rep = ctx.profile_report()
    print(key, " ".join("%s=%.3f" % (k, v["ms"] / reps) for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"])))
    ctx.close()
