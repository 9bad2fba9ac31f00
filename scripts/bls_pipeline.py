"""The GPU side of the reference's BLS example at n = 128 (src/bin/bls_aggregation.rs), timed end to end on one MI355X:

    messages -> ms (map + cofactor)                          sipp_map_to_g2
    a = pks + [-G1], b = ms + [agg]: inner_product == 1      sipp_inner_product           (the aggregate verifies)
    sipp_prove_native / sipp_verify_native                   native chain -> obligation lists
    six STARK proofs, concurrently on six ctxs:              MapToG2, the cofactor G2ExpStark (254 obligations), G1 / G2 / Fq12 of SIPP,
                                                             the final pairing pairing_circuit(final_A, final_B) == final_Z (:76-77)

Keys and signatures are made on the host with Python big integers (the signers' side, not timed).  usage: bls_pipeline.py [n=128] [reps=3]
(by hand, on a GPU box; the proofs are checked by the oracle's verifier when SIPP_BLS_VERIFY=1)."""
import os, sys, time, random
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle", "py"))
import sipp_amd, bn254 as bn

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rnd = random.Random(0xb15)
L = sipp_amd.lib()
sks = [rnd.randrange(1, bn.R) for _ in range(n - 1)]
pks = [bn.g1_mul(bn.G1, sk) for sk in sks]
msgs = [(rnd.randrange(bn.P), rnd.randrange(bn.P)) for _ in range(n - 1)]
words = np.array([bn.fq_to_u32(u[0]) + bn.fq_to_u32(u[1]) for u in msgs], dtype=np.uint32)
nio = (n - 1, n - 1, 2 * (n.bit_length() - 1))
kinds = (3, 1, 0, 1, 2, 6)         # MapToG2, cofactor (G2), the SIPP instance's G1 / G2 / Fq12, the final pairing
sizes = (n - 1, 2 * (n - 1)) + nio + (1,)
ctxs = [sipp_amd.Ctx(workspace_bytes=max(1 << 30, L.sipp_workspace_bytes(k, m))) for k, m in zip(kinds, sizes)]
for c, lvl in zip(ctxs, (1, 0, -1, 0, 1, 1)):
    c._ck(L.sipp_ctx_set_stream_priority(c.h, lvl), "prio")
main = ctxs[3]                      # the SIPP G2 ctx also runs the native steps
t0 = time.perf_counter()
map_recs, cof_recs, ms_words = main.map_to_g2(words)
ms = [((bn.u32_to_fq(list(w[0:8])), bn.u32_to_fq(list(w[8:16]))), (bn.u32_to_fq(list(w[16:24])), bn.u32_to_fq(list(w[24:32])))) for w in ms_words]
agg = None
for m, sk in zip(ms, sks):
    agg = bn.g2_add(agg, bn.g2_mul(m, sk))
print("host side (keys, %d signatures, aggregate): %.1f s" % (n - 1, time.perf_counter() - t0), flush=True)
A = np.array([bn.g1_to_u32(p) for p in pks + [bn.g1_neg(bn.G1)]], dtype=np.uint32)
one = np.zeros(96, dtype=np.uint32); one[0] = 1

def run():
    t = [time.perf_counter()]
    recs, cof, msw = main.map_to_g2(words);                          t.append(time.perf_counter())
    B = np.concatenate([msw, np.array([bn.g2_to_u32(agg)], dtype=np.uint32)])
    ok = bool((main.inner_products(A, B)[0] == one).all());          t.append(time.perf_counter())
    # the two proofs that need only the messages start now, beside the native chain
    ctxs[0].prove_async(3, recs); ctxs[1].prove_async(1, cof)
    proof = main.prove_native(A, B)
    okv, st, ios = main.verify_native(A, B, proof);                  t.append(time.perf_counter())
    for c, k, a in zip(ctxs[2:5], kinds[2:5], ios):
        c.prove_async(k, a)
    ctxs[5].prove_async(6, np.ascontiguousarray(st[-144:].reshape(1, 144)))      # (final_A, final_B, final_Z) of the statement
    pfs = [c.wait() for c in ctxs];                                  t.append(time.perf_counter())
    return ok and okv, pfs, [1e3 * (b - a) for a, b in zip(t, t[1:])], 1e3 * (t[-1] - t[0])

run()
for _ in range(reps):
    ok, pfs, ph, total = run()
    print("aggregate verifies: %s | map+cofactor %.1f | pairing product %.1f | native chain %.1f | proofs (rest) %.1f | total %.1f ms  (proof words %s)"
          % (ok, ph[0], ph[1], ph[2], ph[3], total, [len(p) for p in pfs]), flush=True)
if os.environ.get("SIPP_BLS_VERIFY"):
    from tests import _oracle
    print("oracle verifier:", [_oracle.stark_verify(p) for p in pfs])
