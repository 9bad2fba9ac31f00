"""Pinning parity against the reference ITSELF (SURVEY.md section 8c "what would pin parity").

`rust_shim/examples/capture_golden.rs` (source only: no Rust toolchain here) dumps, with the reference's own crates, the values of
the seeded n = 128 input set: native SIPP proof, challenges, statement limbs, Poseidon sponge / compression outputs, a
PolynomialBatch cap and the three starky proofs.  This file is the consumer: when `tests/golden/reference/capture_n<N>.json` exists
it is compared SECTION BY SECTION with the CPU oracle (and with the HIP path in the GPU test) and the first differing section is
named; without a capture the comparison is skipped with that reason.  The comparator itself is exercised on every run with a
capture-shaped document made from the oracle's own values (n = 4)."""
import glob
import json
import os

import numpy as np
import pytest

from tests import _oracle
from oracle.py import bn254 as bn
from oracle.py import sipp_native as sn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEEDS = {4: 7, 8: 0x51515050, 128: 0x51515050 + 1, 1024: 0x51515050 + 2}
BATCH_SEED = 0xba7c4


def upstream_rules_cfg():
    cfg = _oracle.default_config()
    cfg.fs_rule, cfg.lookup_rule, cfg.pow_rule = 1, 1, 1
    return cfg


def hash_inputs(k):
    return [(0x0123456789abcdef * (i + 1) % (1 << 64)) % _oracle.P for i in range(k)]


def batch_values():
    g = sn.splitmix64(BATCH_SEED)
    return np.array([[next(g) % _oracle.P for _ in range(1024)] for _ in range(4)], dtype=np.uint64)


def proof_sections(flat):
    """the sections of a flat proof a capture carries (oracle/stark.c header comment): caps, openings, final polynomial, pow witness"""
    pf = [int(x) for x in flat]
    log_n, W, P, Q, cap_h, rounds, final_len, ppi, nio = pf[2], pf[4], pf[5], pf[6], pf[7], pf[8], pf[9], pf[11], pf[3]
    cap = 4 << cap_h
    pos = 16
    out = {"degree_bits": log_n}
    for name in ("trace_cap", "permutation_zs_cap", "quotient_polys_cap"):
        out[name] = pf[pos:pos + cap]
        pos += cap
    for name, cnt in (("local_values", W), ("next_values", W), ("permutation_zs", P), ("permutation_zs_next", P), ("quotient_polys", Q)):
        out[name] = pf[pos:pos + 2 * cnt]
        pos += 2 * cnt
    out["commit_phase_merkle_caps"] = []
    for r in range(rounds):
        leaves_log = log_n + pf[13] - pf[14] * (r + 1)
        c = 4 << min(cap_h, leaves_log)
        out["commit_phase_merkle_caps"].append(pf[pos:pos + c])
        pos += c
    out["final_poly"] = pf[pos:pos + 2 * final_len]
    pos += 2 * final_len
    out["pow_witness"] = pf[pos]
    out["public_inputs"] = pf[len(pf) - nio * ppi:]
    return out


def flatten(v):
    if isinstance(v, (list, tuple)):
        return [x for e in v for x in flatten(e)]
    return [int(v)]


def oracle_document(n, stark_proofs=None):
    """what the oracle computes for every section of the schema (a capture-shaped dict)"""
    seed = SEEDS.get(n, n)
    A, B = sn.synthetic_inputs(n, seed)
    path = os.path.join(ROOT, "tests", "golden", "sipp_n%d_ios.npz" % n)
    d = np.load(path)
    f12 = d["fq12"]
    rounds = n.bit_length() - 1
    # proof list in the reference's (reversed) order: prover_native.rs:78 -- [Z_R(last round), Z_L(last), ..., Z_R(1), Z_L(1), Z]
    msgs = [f12[0, 96:192]]
    for r in range(rounds):
        msgs += [f12[2 * r, 0:96], f12[2 * r + 1, 0:96]]
    doc = {
        "schema": "sipp-capture-1", "n": n, "seed": seed,
        "inputs": {"A": [bn.g1_to_u32(p) for p in A], "B": [bn.g2_to_u32(q) for q in B]},
        "native": {"proof": [[int(x) for x in m] for m in reversed(msgs)],
                   "challenges": [[int(x) for x in f12[2 * r, 192:200]] for r in range(rounds)],
                   "statement": [int(x) for x in d["statement"]]},
        "poseidon": {"hash_no_pad": [{"input": hash_inputs(k), "output": [int(x) for x in _oracle.hash_no_pad(hash_inputs(k))]}
                                     for k in (0, 1, 4, 7, 8, 9, 16, 21)]},
        "polynomial_batch": {"log_n": 10, "ncols": 4, "rate_bits": 1, "cap_height": 4, "seed": BATCH_SEED,
                             "cap": [[int(x) for x in row] for row in _oracle.Batch(batch_values(), 10).cap]},
    }
    l, r = _oracle.hash_no_pad(hash_inputs(5)), _oracle.hash_no_pad(hash_inputs(6))
    doc["poseidon"]["two_to_one"] = [{"left": [int(x) for x in l], "right": [int(x) for x in r],
                                      "output": [int(x) for x in _oracle.two_to_one(l, r)]}]
    if stark_proofs is not None:
        doc["stark"] = {k: proof_sections(pf) for k, pf in zip(("g1", "g2", "fq12"), stark_proofs)}
    return doc


SECTION_ORDER = [("inputs", "A"), ("inputs", "B"), ("native", "proof"), ("native", "challenges"), ("native", "statement"),
                 ("poseidon", "hash_no_pad"), ("poseidon", "two_to_one"), ("polynomial_batch", "cap")]
STARK_ORDER = ["degree_bits", "public_inputs", "trace_cap", "permutation_zs_cap", "local_values", "next_values", "permutation_zs",
               "permutation_zs_next", "quotient_polys_cap", "quotient_polys", "commit_phase_merkle_caps", "final_poly", "pow_witness"]


def first_difference(capture, ours):
    """None, or 'section.subsection' of the first part (in protocol order) where the two documents differ"""
    if capture.get("schema") != "sipp-capture-1" or capture.get("n") != ours["n"] or capture.get("seed") != ours["seed"]:
        return "schema / n / seed"
    for sec, sub in SECTION_ORDER:
        a, b = capture[sec][sub], ours[sec][sub]
        if sec == "poseidon":
            a = [(flatten(c.get("input", [c.get("left"), c.get("right")])), flatten(c["output"])) for c in a]
            b = [(flatten(c.get("input", [c.get("left"), c.get("right")])), flatten(c["output"])) for c in b]
            if a != b:
                return "%s.%s" % (sec, sub)
        elif flatten(a) != flatten(b):
            return "%s.%s" % (sec, sub)
    if "stark" in ours and isinstance(capture.get("stark"), dict) and "todo" not in capture["stark"]:
        for kind in ("g1", "g2", "fq12"):
            for sub in STARK_ORDER:
                if flatten(capture["stark"][kind][sub]) != flatten(ours["stark"][kind][sub]):
                    return "stark.%s.%s" % (kind, sub)
    return None


def captures():
    return sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "reference", "capture_n*.json")))


def test_comparator_accepts_the_oracles_own_document_and_names_what_differs():
    cfg = upstream_rules_cfg()
    d = np.load(os.path.join(ROOT, "tests", "golden", "sipp_n4_ios.npz"))
    proofs = [_oracle.stark_prove(k, d[key], cfg) for k, key in ((0, "g1"), (1, "g2"), (2, "fq12"))]
    ours = oracle_document(4, proofs)
    cap = json.loads(json.dumps(ours))                       # what a capture file of the same values would hold
    assert first_difference(cap, ours) is None
    # the native proof list ends with Z = prod e(A_i, B_i) and the challenges are the exponents of the Fq12 obligations
    A, B = sn.synthetic_inputs(4, 7)
    assert cap["native"]["proof"][-1] == bn.f12_to_u32(bn.multi_pairing(A, B))
    t = json.loads(json.dumps(ours)); t["native"]["challenges"][1][0] ^= 1
    assert first_difference(t, ours) == "native.challenges"
    t = json.loads(json.dumps(ours)); t["poseidon"]["hash_no_pad"][3]["output"][2] += 1
    assert first_difference(t, ours) == "poseidon.hash_no_pad"
    t = json.loads(json.dumps(ours)); t["polynomial_batch"]["cap"][5][0] += 1
    assert first_difference(t, ours) == "polynomial_batch.cap"
    t = json.loads(json.dumps(ours)); t["stark"]["g2"]["permutation_zs_cap"][0] += 1
    assert first_difference(t, ours) == "stark.g2.permutation_zs_cap"
    t = json.loads(json.dumps(ours)); t["stark"]["fq12"]["pow_witness"] += 1
    assert first_difference(t, ours) == "stark.fq12.pow_witness"
    t = json.loads(json.dumps(ours)); t["stark"] = {"todo": "not serialised"}
    assert first_difference(t, ours) is None                 # a capture without the STARK part is compared up to there


def test_upstream_rule_digests_are_committed():
    """the configuration closest to upstream (plain kinds, fs_rule = lookup_rule = pow_rule = 1) has committed
    n = 128 digests next to the default ones (tools/gen_golden.py digests128)"""
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "proof_digests_n128.json")))
    for key in ("g1_upstream_rules", "g2_upstream_rules", "fq12_upstream_rules"):
        assert len(gold[key]["sha256"]) == 64 and gold[key]["kind"] in (0, 1, 2)
        assert gold[key]["sha256"] != gold[key.replace("_upstream_rules", "")]["sha256"]


def test_oracle_against_the_reference_capture():
    files = captures()
    if not files:
        pytest.skip("no tests/golden/reference/capture_n*.json: the reference cannot be built in this image (no cargo); "
                    "rust_shim/examples/capture_golden.rs writes one where it can")
    for path in files:
        cap = json.load(open(path))
        n = cap["n"]
        cfg = upstream_rules_cfg()
        d = np.load(os.path.join(ROOT, "tests", "golden", "sipp_n%d_ios.npz" % n))
        proofs = None
        if isinstance(cap.get("stark"), dict) and "todo" not in cap["stark"]:
            proofs = [_oracle.stark_prove(k, d[key], cfg) for k, key in ((0, "g1"), (1, "g2"), (2, "fq12"))]
        diff = first_difference(cap, oracle_document(n, proofs))
        assert diff is None, "%s: the oracle first differs from the reference in section %s" % (os.path.basename(path), diff)


@pytest.mark.gpu
def test_hip_against_the_reference_capture():
    files = captures()
    if not files:
        pytest.skip("no tests/golden/reference/capture_n*.json (see test_oracle_against_the_reference_capture)")
    import sipp_amd
    for path in files:
        cap = json.load(open(path))
        n = cap["n"]
        A = np.array(cap["inputs"]["A"], dtype=np.uint32)
        B = np.array(cap["inputs"]["B"], dtype=np.uint32)
        cfg = sipp_amd.default_config()
        cfg.fs_rule, cfg.lookup_rule, cfg.pow_rule = 1, 1, 1
        L = sipp_amd.lib()
        ctx = sipp_amd.Ctx(workspace_bytes=max(L.sipp_workspace_bytes(k, n) for k in (0, 1, 2)), cfg=cfg)
        try:
            proof = ctx.prove_native(A, B)
            assert [list(map(int, m)) for m in proof.reshape(-1, 96)] == cap["native"]["proof"], "native.proof"
            ok, st, ios = ctx.verify_native(A, B, proof)
            assert ok and [int(x) for x in st] == cap["native"]["statement"], "native.statement"
            if isinstance(cap.get("stark"), dict) and "todo" not in cap["stark"]:
                for k, key in ((0, "g1"), (1, "g2"), (2, "fq12")):
                    ours = proof_sections(ctx.prove(k, ios[k]))
                    for sub in STARK_ORDER:
                        assert flatten(cap["stark"][key][sub]) == flatten(ours[sub]), "stark.%s.%s" % (key, sub)
        finally:
            ctx.close()
