"""Quick per-kernel timing of the commitment pipeline at n=128-like shapes (GPU box)."""
import sys, json
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sipp_amd
from sipp_amd._lib import to_device

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ncols = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
ctx = sipp_amd.Ctx(workspace_bytes=4 << 30)
g = torch.Generator(device="cuda"); g.manual_seed(1)
vals = torch.randint(0, 2**62, (ncols, 1 << log_n), dtype=torch.int64, device="cuda", generator=g)
coeffs = torch.empty_like(vals)
lde = torch.empty((ncols, 2 << log_n), dtype=torch.int64, device="cuda")
tree = torch.empty((4 << log_n, 4), dtype=torch.int64, device="cuda")
for it in range(2):
    ctx.commit(vals, log_n, (coeffs, lde, tree))
ctx.profile(True); ctx.profile_reset()
K = 5
ctx.timer_start()
for it in range(K):
    ctx.commit(vals, log_n, (coeffs, lde, tree))
ms = ctx.timer_stop()
rep = ctx.profile_report()
print("commit N=2^%d W=%d: %.3f ms/iter" % (log_n, ncols, ms / K))
n = 1 << log_n
for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"]):
    print("  %-18s calls/iter %4d  %.3f ms/iter" % (k, v["calls"] // K, v["ms"] / K))
perms = (2 * n) * ((ncols + 7) // 8)
leaf = [k for k in ("poseidon_leaves", "poseidon_leaves_pair") if k in rep][0]
print("leaf perms/s (%s): %.3f G" % (leaf, perms / (rep[leaf]["ms"] / K * 1e-3) / 1e9))
ntt_bytes = 8 * n * ncols
print("lde bytes (1 read + 2 write): %.1f MB" % (3 * ntt_bytes / 1e6))
