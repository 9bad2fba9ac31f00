#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/ with the CPU restatements (oracle/py):
native SIPP obligations (STARK IO records) for seeded synthetic inputs A_i = [s_i]G1, B_i = [t_i]G2
(SplitMix64).  SELF-GOLDEN: produced by this repository's restatement of
reference src/prover_native.rs / src/verifier_native.rs / src/transcript_native.rs, not by the reference
binary (no cargo in this image).

    python tools/gen_golden.py 4 8 128
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.py import sipp_native as sn  # noqa: E402

SEEDS = {4: 7, 8: 0x51515050, 128: 0x51515050 + 1, 1024: 0x51515050 + 2}


def proof_digests():
    """SELF-GOLDEN digests of the CPU oracle's three sub-proofs for the n = 4 fixture (sha256 over the u64 words):
    pins the oracle -- AIR tables, lookup fill rule, Fiat-Shamir order, FRI -- against silent drift.  The GPU path is
    compared with the oracle word for word in tests/test_gpu_stark.py, so these digests pin it too."""
    import hashlib
    import json
    from tests import _oracle
    d = np.load(os.path.join(ROOT, "tests", "golden", "sipp_n4_ios.npz"))
    out = {}
    for kind, key in ((0, "g1"), (1, "g2"), (2, "fq12")):
        pf = _oracle.stark_prove(kind, d[key])
        out[key] = {"words": int(len(pf)), "log_n": int(pf[2]), "W": int(pf[4]), "P": int(pf[5]),
                    "pow_witness_index": None, "sha256": hashlib.sha256(pf.tobytes()).hexdigest()}
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "proof_digests_n4.json"), "w"), indent=1)
    print("proof digests:", {k: v["sha256"][:16] for k, v in out.items()})


def proof_digests_n128():
    """Digests of the CPU oracle's three sub-proofs at the BASELINE size n = 128 (about ten minutes of CPU on 8 cores):
    tests/test_gpu_stark.py compares the GPU proofs' digests with them, i.e. full-size word-for-word parity without
    running the CPU prover inside the test."""
    import hashlib
    import json
    from tests import _oracle
    d = np.load(os.path.join(ROOT, "tests", "golden", "sipp_n128_ios.npz"))
    out = {}
    for kind, key in ((0, "g1"), (1, "g2"), (2, "fq12")):
        t = time.time()
        pf = _oracle.stark_prove(kind, d[key])
        assert _oracle.stark_verify(pf) == 0
        out[key] = {"words": int(len(pf)), "log_n": int(pf[2]), "W": int(pf[4]), "P": int(pf[5]),
                    "sha256": hashlib.sha256(pf.tobytes()).hexdigest(), "oracle_seconds": round(time.time() - t, 1)}
        print(key, out[key], flush=True)
        json.dump(out, open(os.path.join(ROOT, "tests", "golden", "proof_digests_n128.json"), "w"), indent=1)


def main():
    if sys.argv[1:] == ["digests"]:
        return proof_digests()
    if sys.argv[1:] == ["digests128"]:
        return proof_digests_n128()
    for n in [int(x) for x in sys.argv[1:]]:
        t = time.time()
        A, B = sn.synthetic_inputs(n, SEEDS.get(n, n))
        proof = sn.sipp_prove_native(A, B)
        ok, st, obl = sn.sipp_verify_native(A, B, proof)
        assert ok, "native SIPP verification failed"
        g1, g2, f12 = sn.io_records(obl)
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", "sipp_n%d_ios.npz" % n), g1=g1, g2=g2, fq12=f12,
                            statement=np.array(sn.statement_to_u32(st), dtype=np.uint32))
        print("n=%d: %s %s %s in %.1fs" % (n, g1.shape, g2.shape, f12.shape, time.time() - t))


if __name__ == "__main__":
    main()
