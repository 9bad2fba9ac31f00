#!/usr/bin/env python3
"""One-off word-for-word check of whole proofs at trace lengths the unit tests do not reach (GPU box; the oracle side takes
minutes): records taken from the n = 1024 fixture.  usage: stress_sizes.py "kind:count,kind:count,..." """
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sipp_amd  # noqa: E402
from tests import _oracle  # noqa: E402

os.environ.setdefault("OMP_NUM_THREADS", "16")
d = np.load(os.path.join(ROOT, "tests", "golden", "sipp_n1024_ios.npz"))
lists = [d["g1"], d["g2"], d["fq12"]]
cases = [(int(a), int(b)) for a, b in (c.split(":") for c in (sys.argv[1] if len(sys.argv) > 1 else "0:40,2:20,1:40,0:130").split(","))]
bad = 0
for kind, num in cases:
    ios = np.ascontiguousarray(lists[kind & 3][:num])     # kinds 4 / 5: the hardened G1 / G2 AIRs over the same records
    ctx = sipp_amd.Ctx(workspace_bytes=sipp_amd.lib().sipp_workspace_bytes(kind, num))
    try:
        t = time.time()
        got = ctx.prove(kind, ios).copy()
        tg = time.time() - t
    finally:
        ctx.close()
    t = time.time()
    ref = _oracle.stark_prove(kind, ios)
    to = time.time() - t
    ok = len(got) == len(ref) and bool((got == ref).all())
    bad += not ok
    print("%s kind %d records %d: log_n %d, %d words, GPU %.2f s, oracle %.0f s" % ("ok  " if ok else "MISMATCH", kind, num, int(got[2]), len(got), tg, to),
          flush=True)
sys.exit(1 if bad else 0)
