"""MapToG2 (kind 3) on a GPU box: proofs for message counts around the padding boundaries, word for word against the oracle's
(by hand; the oracle needs ~5 - 20 s per proof).  usage: stress_mapg2.py [counts=1,2,3,128,129,257]"""
import os, sys, time, random
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle", "py"))
import sipp_amd, bn254
from tests import _oracle

counts = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,2,3,128,129,257").split(",")]
rnd = random.Random(99)
L = sipp_amd.lib()
ctx = sipp_amd.Ctx(workspace_bytes=L.sipp_workspace_bytes(3, max(counts)))
bad = 0
for n in counts:
    words = np.array([bn254.fq_to_u32(rnd.randrange(bn254.P)) + bn254.fq_to_u32(rnd.randrange(bn254.P)) for _ in range(n)], dtype=np.uint32)
    recs = ctx.map_to_g2(words, cofactor=False)
    t0 = time.perf_counter(); got = ctx.prove(3, recs); t1 = time.perf_counter()
    want = _oracle.stark_prove(3, recs); t2 = time.perf_counter()
    ok = got.shape == want.shape and bool((got == want).all()) and _oracle.stark_verify(got) == 0
    bad += not ok
    print("messages %4d  shape %s  gpu %.1f ms  oracle %.1f s  %s" % (n, ctx.shape(3, n), 1e3 * (t1 - t0), t2 - t1, "ok" if ok else "MISMATCH"), flush=True)
sys.exit(1 if bad else 0)
