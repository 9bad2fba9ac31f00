"""Timing of the final-pairing STARK (kind 6: pairing_circuit(final_A, final_B) == final_Z, reference src/bin/bls_aggregation.rs:76-77)
on one MI355X: `records` pairings in one proof.  usage: perf_pairing_stark.py [records=1] [reps=5]  (GPU box; SIPP_VERIFY=1 runs the
oracle's verifier over the last proof)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sipp_amd

records = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
st = np.load(os.path.join(ROOT, "tests", "golden", "sipp_n128_ios.npz"))["statement"]
rec = np.ascontiguousarray(np.repeat(st[-144:].reshape(1, 144), records, axis=0))
ctx = sipp_amd.Ctx(workspace_bytes=sipp_amd.lib().sipp_workspace_bytes(6, records))
ctx.prove(6, rec)
ctx.profile(True); ctx.profile_reset()
t = time.perf_counter()
for _ in range(reps):
    pf = ctx.prove(6, rec)
ms = 1e3 * (time.perf_counter() - t) / reps
print("final pairing STARK: %d record(s), shape %s, %.2f ms per proof, %d proof words" % (records, ctx.shape(6, records), ms, len(pf)))
for k, v in sorted(ctx.profile_report().items(), key=lambda kv: -kv[1]["ms"])[:14]:
    print("  %-22s calls/proof %4d  %.3f ms/proof" % (k, v["calls"] // reps, v["ms"] / reps))
t = time.perf_counter()
out = ctx.exp_outputs(6, rec)
print("value alone (sipp_exp_outputs): %.2f ms" % (1e3 * (time.perf_counter() - t)))
if os.environ.get("SIPP_VERIFY"):
    from tests import _oracle
    print("oracle verifier:", _oracle.stark_verify(pf))
