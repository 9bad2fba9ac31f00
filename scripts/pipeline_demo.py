"""End to end on one GPU: points A, B -> sipp_prove_native -> sipp_verify_native (statement + obligation lists) ->
sipp_instance_prove (three STARK proofs).  The flow of the reference's test_sipp_circuit without the outer plonky2 proof."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sipp_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
st = np.load(os.path.join(ROOT, "tests", "golden", "sipp_n%d_ios.npz" % n))["statement"]
A, B = st[: 16 * n].reshape(n, 16), st[16 * n: 48 * n].reshape(n, 32)
lg = n.bit_length() - 1
inst = sipp_amd.Instance([n - 1, n - 1, 2 * lg])
nctx = inst.ctxs[1]
for it in range(3):
    t0 = time.perf_counter(); proof = nctx.prove_native(A, B)
    t1 = time.perf_counter(); ok, stm, ios = nctx.verify_native(A, B, proof)
    t2 = time.perf_counter(); proofs = inst.prove(ios)
    t3 = time.perf_counter()
    print("n=%d pass %d: prove_native %.1f ms, verify_native %.1f ms, three STARK proofs %.1f ms, total %.1f ms (accepted %s, %d proof words)"
          % (n, it, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t3 - t0), ok, sum(len(p) for p in proofs)))
inst.close()
