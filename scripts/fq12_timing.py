"""Phases and kernels of ONE Fq12ExpStark proof of the n = 128 instance, alone on the GPU (SIPP_HOST_TIMING=1 prints the host-side phase
times): the lone instance's long pole -- 44.7 ms, 30.3 of them the two-lane leaf hashing of 2^14 leaves x (618 + 378) dependent permutations."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, sipp_amd
d = np.load("tests/golden/sipp_n128_ios.npz")
ios = d["fq12"]
ctx = sipp_amd.Ctx(workspace_bytes=sipp_amd.lib().sipp_workspace_bytes(2, ios.shape[0]))
for i in range(3):
    t = time.perf_counter(); ctx.prove(2, ios); print("prove ms", 1e3 * (time.perf_counter() - t), flush=True)
ctx.profile(True); ctx.profile_reset(); ctx.prove(2, ios)
rep = ctx.profile_report()
for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"])[:25]:
    print("    %-22s calls %4d  %8.3f ms" % (k, v["calls"], v["ms"]))
print("sum kernels", sum(v["ms"] for v in rep.values()))
