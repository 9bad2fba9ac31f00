"""MapToG2 on a GPU box: messages -> records (native map + cofactor clearing) -> the MapToG2 proof -> the G2ExpStark proof of the
cofactor obligations; per-kernel breakdown of the MapToG2 proof.  usage: perf_mapg2.py [messages=127]"""
import os, sys, time, random
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "py"))
import sipp_amd, bn254

n = int(sys.argv[1]) if len(sys.argv) > 1 else 127
rnd = random.Random(7)
words = np.array([bn254.fq_to_u32(rnd.randrange(bn254.P)) + bn254.fq_to_u32(rnd.randrange(bn254.P)) for _ in range(n)], dtype=np.uint32)
L = sipp_amd.lib()
ws = max(L.sipp_workspace_bytes(3, n), L.sipp_workspace_bytes(1, 2 * n))
ctx = sipp_amd.Ctx(workspace_bytes=ws)
print("messages %d; shapes: map %s, cofactor G2 exp %s; workspace %.1f GB" % (n, ctx.shape(3, n), ctx.shape(1, 2 * n), ws / 2**30))

def timed(f, k=3):
    f()
    t0 = time.perf_counter()
    for _ in range(k):
        r = f()
    return (time.perf_counter() - t0) / k * 1e3, r

ms, (recs, g2, pts) = timed(lambda: ctx.map_to_g2(words))
print("sipp_map_to_g2 (map, G + [h]Q, - G): %.2f ms" % ms)
ms, _ = timed(lambda: ctx.map_to_g2(words, cofactor=False))
print("  the map alone: %.2f ms" % ms)
ms, proof = timed(lambda: ctx.prove(3, recs))
print("MapToG2 proof: %.2f ms, %d words" % (ms, len(proof)))
ctx.profile(True); ctx.profile_reset()
ctx.prove(3, recs)
rep = ctx.profile_report()
for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"])[:14]:
    print("    %-22s calls %4d  %8.3f ms" % (k, v["calls"], v["ms"]))
ctx.profile(False)
ms, proof2 = timed(lambda: ctx.prove(1, g2), k=2)
print("G2ExpStark proof of the %d cofactor obligations: %.2f ms, %d words" % (2 * n, ms, len(proof2)))
