"""Timing of the native SIPP chain on the GPU (sipp_prove_native / sipp_verify_native) at the fixture sizes."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sipp_amd
for n in [int(x) for x in (sys.argv[1:] or ["128"])]:
    d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "sipp_n%d_ios.npz" % n))
    st = d["statement"]
    A, B = st[: 16 * n].reshape(n, 16), st[16 * n: 48 * n].reshape(n, 32)
    ctx = sipp_amd.Ctx(workspace_bytes=max(1 << 30, sipp_amd.lib().sipp_workspace_bytes(1, max(1, n // 2))))
    ctx.prove_native(A, B)
    ctx.profile(True); ctx.profile_reset()
    t = time.perf_counter(); proof = ctx.prove_native(A, B); tp = time.perf_counter() - t
    rep = ctx.profile_report()
    t = time.perf_counter(); ok, _, _ = ctx.verify_native(A, B, proof); tv = time.perf_counter() - t
    print("n=%d: prove_native %.1f ms (%d pairings -> %.0f pairings/s), verify_native %.1f ms, accepted %s" % (n, 1e3 * tp, 3 * n - 2, (3 * n - 2) / tp, 1e3 * tv, ok))
    for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"])[:8]:
        print("   %-24s calls %3d  %.2f ms" % (k, v["calls"], v["ms"]))
    ctx.close()
