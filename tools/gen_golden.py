#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/ with the CPU restatements (oracle/py):
native SIPP obligations (STARK IO records) for seeded synthetic inputs A_i = [s_i]G1, B_i = [t_i]G2
(SplitMix64).  SELF-GOLDEN: produced by this repository's restatement of
reference src/prover_native.rs / src/verifier_native.rs / src/transcript_native.rs, not by the reference
binary (no cargo in this image).

    python tools/gen_golden.py 4 8 128
    python tools/gen_golden.py mapg2        # tests/golden/mapg2_vectors.json
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
    path = os.path.join(ROOT, "tests", "golden", "proof_digests_n128.json")
    if os.path.exists(path):
        out = json.load(open(path))        # digests128 <key> ...: only the named proofs are recomputed
    todo = sys.argv[2:] or ["g1", "g2", "fq12", "g1_hardened", "g2_hardened", "g1_upstream_rules", "g2_upstream_rules", "fq12_upstream_rules"]
    # the hardened G1 / G2 AIRs (API kinds 4 / 5, the variant bench.py's headline proves since round 4) over the same IO records;
    # *_upstream_rules: the plain kinds under fs_rule = lookup_rule = pow_rule = 1, the configuration closest to upstream's starky as
    # recalled -- what tests/test_reference_capture.py compares a cargo capture with
    up = _oracle.default_config()
    up.fs_rule, up.lookup_rule, up.pow_rule = 1, 1, 1
    for kind, key, src, cfg in ((0, "g1", "g1", None), (1, "g2", "g2", None), (2, "fq12", "fq12", None), (4, "g1_hardened", "g1", None),
                                (5, "g2_hardened", "g2", None), (0, "g1_upstream_rules", "g1", up), (1, "g2_upstream_rules", "g2", up),
                                (2, "fq12_upstream_rules", "fq12", up)):
        if key not in todo:
            continue
        t = time.time()
        pf = _oracle.stark_prove(kind, d[src], cfg)
        assert _oracle.stark_verify(pf, cfg) == 0
        out[key] = {"kind": kind, "words": int(len(pf)), "log_n": int(pf[2]), "W": int(pf[4]), "P": int(pf[5]),
                    "sha256": hashlib.sha256(pf.tobytes()).hexdigest(), "oracle_seconds": round(time.time() - t, 1)}
        print(key, out[key], flush=True)
        json.dump(out, open(os.path.join(ROOT, "tests", "golden", "proof_digests_n128.json"), "w"), indent=1)


def proof_digests_large():
    """word-for-word parity at the LARGE configs without running the CPU prover inside the tests -- sha256 of the
    oracle's proofs for BASELINE configs[2] (n = 1024 on one GPU: hardened G1 / G2, Fq12) and for rank 3's world-8 shard of
    configs[4] (n = 4096: records [3 n / 8, 4 n / 8) of every list, sipp_io_shard).  Hours of CPU and ~40 GB of memory for the
    n = 1024 G2 proof: run one key at a time (`digests_large n1024.g2_hardened`)."""
    import hashlib
    import json
    from tests import _oracle
    path = os.path.join(ROOT, "tests", "golden", "proof_digests_large.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    jobs = {}
    d = np.load(os.path.join(ROOT, "tests", "golden", "sipp_n1024_ios.npz"))
    for kind, key, src in ((2, "fq12", "fq12"), (4, "g1_hardened", "g1"), (5, "g2_hardened", "g2")):
        jobs["n1024." + key] = (kind, d[src])
    d = np.load(os.path.join(ROOT, "tests", "golden", "sipp_n4096_ios.npz"))
    for kind, key, src in ((2, "fq12", "fq12"), (4, "g1_hardened", "g1"), (5, "g2_hardened", "g2")):
        a = d[src]
        first, last = a.shape[0] * 3 // 8, a.shape[0] * 4 // 8        # sipp_io_shard(num_io, 8, 3)
        jobs["n4096_world8_rank3." + key] = (kind, a[first:last])
    for name in sys.argv[2:] or sorted(jobs):
        kind, recs = jobs[name]
        t = time.time()
        pf = _oracle.stark_prove(kind, np.ascontiguousarray(recs))
        assert _oracle.stark_verify(pf) == 0
        out[name] = {"kind": kind, "records": int(recs.shape[0]), "words": int(len(pf)), "log_n": int(pf[2]), "W": int(pf[4]), "P": int(pf[5]),
                     "sha256": hashlib.sha256(pf.tobytes()).hexdigest(), "oracle_seconds": round(time.time() - t, 1)}
        print(name, out[name], flush=True)
        json.dump(out, open(path, "w"), indent=1, sort_keys=True)


def mapg2_fixture():
    """SELF-GOLDEN vectors of the messages -> G2 step (HISTORY.md section 7b): 12 messages (seeded, plus u = 0, u in Fp, u = c u and
    the inv0 messages u^2 g(Z) = +-1) with their images under the Python reading of the map (oracle/py/map_to_g2.py), the
    cofactor-cleared points, and the sha256 of the CPU oracle's MapToG2 proof of the 12 records."""
    import hashlib
    import json
    import random
    sys.path.insert(0, os.path.join(ROOT, "oracle", "py"))
    import bn254
    import map_to_g2 as M
    from tests import _oracle
    rnd = random.Random(0x6d617067)
    us = [(rnd.randrange(bn254.P), rnd.randrange(bn254.P)) for _ in range(6)] + [(0, 0), (12345, 0), (0, 67890)]
    r = M.sqrt_even(bn254.f2_inv(M.C1)) or M.sqrt_even(bn254.f2_neg(bn254.f2_inv(M.C1)))
    us += [r, bn254.f2_neg(r), (bn254.P - 1, bn254.P - 1)]
    recs, cleared, branch = [], [], []
    for u in us:
        w = M.witness(u)
        q = (w["XS"], w["Y"])
        recs.append(bn254.fq_to_u32(u[0]) + bn254.fq_to_u32(u[1]) + bn254.g2_to_u32(q))
        cleared.append(bn254.g2_to_u32(bn254.g2_mul(q, bn254.G2_COFACTOR)))
        branch.append([w["e1"], w["e2"], w["z"]])
    pf = _oracle.stark_prove(3, np.array(recs, dtype=np.uint32))
    assert _oracle.stark_verify(pf) == 0
    out = {"records": recs, "cleared": cleared, "branch_e1_e2_z": branch,
           "proof": {"words": int(len(pf)), "log_n": int(pf[2]), "W": int(pf[4]), "P": int(pf[5]), "sha256": hashlib.sha256(pf.tobytes()).hexdigest()}}
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "mapg2_vectors.json"), "w"), indent=0)
    print("mapg2 fixture:", len(us), "messages, branches", sorted(set(map(tuple, branch))), "proof", out["proof"]["sha256"][:16])


def main():
    if sys.argv[1:] == ["mapg2"]:
        return mapg2_fixture()
    if sys.argv[1:] == ["digests"]:
        return proof_digests()
    if sys.argv[1:2] == ["digests128"]:
        return proof_digests_n128()
    if sys.argv[1:2] == ["digests_large"]:
        return proof_digests_large()
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
