#!/usr/bin/env python3
"""One-off lifecycle stress (GPU box): (A) six ctxs created and used from six threads at once; (B) 30 create / prove / destroy
cycles with the device's free memory watched; (C) failing and succeeding proofs alternating on one ctx -- the proofs must stay
word for word the same throughout."""
import os
import sys
import threading

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sipp_amd  # noqa: E402

d = np.load(os.path.join(ROOT, "tests", "golden", "sipp_n4_ios.npz"))
ios = [d["g1"], d["g2"], d["fq12"]]
ref_ctx = sipp_amd.Ctx(workspace_bytes=4 << 30)
want = [ref_ctx.prove(k, ios[k]).copy() for k in range(3)]
ref_ctx.close()
bad = 0

# (A) concurrent creation and use
errs = []


def worker(i):
    try:
        c = sipp_amd.Ctx(workspace_bytes=4 << 30)
        for rep in range(3):
            k = (i + rep) % 3
            got = c.prove(k, ios[k])
            if not (len(got) == len(want[k]) and (got == want[k]).all()):
                errs.append("thread %d rep %d kind %d differs" % (i, rep, k))
        c.close()
    except Exception as e:      # noqa: BLE001
        errs.append("thread %d: %r" % (i, e))


ts = [threading.Thread(target=worker, args=(i,)) for i in range(6)]
[t.start() for t in ts]
[t.join() for t in ts]
print("(A) six threads:", "ok" if not errs else errs)
bad += len(errs)

# (B) create / prove / destroy cycles
torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]
lows = []
for it in range(30):
    c = sipp_amd.Ctx(workspace_bytes=2 << 30)
    got = c.prove(it % 3, ios[it % 3])
    if not (got == want[it % 3]).all():
        bad += 1
        print("(B) cycle %d differs" % it)
    c.close()
    lows.append(torch.cuda.mem_get_info()[0])
leak = free0 - lows[-1]
print("(B) 30 cycles: free memory %.1f -> %.1f MiB (drift %.1f MiB)" % (free0 / 2**20, lows[-1] / 2**20, leak / 2**20))
if leak > (64 << 20):
    bad += 1
    print("(B) LEAK suspected")

# (C) failing and succeeding proofs on one ctx
c = sipp_amd.Ctx(workspace_bytes=4 << 30)
for it in range(20):
    k = it % 3
    if it % 2 == 0:
        broken = ios[k].copy()
        broken[it % broken.shape[0], -1] ^= 1          # a wrong claimed output
        try:
            c.prove(k, broken)
            bad += 1
            print("(C) bad record %d proved" % it)
        except sipp_amd.SippError:
            pass
    else:
        got = c.prove(k, ios[k])
        if not (len(got) == len(want[k]) and (got == want[k]).all()):
            bad += 1
            print("(C) proof %d differs after a failure" % it)
c.close()
print("(C) alternating failures: done")
print("lifecycle stress:", "ok" if bad == 0 else "%d problems" % bad)
sys.exit(1 if bad else 0)
