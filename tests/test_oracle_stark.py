"""CPU oracle STARK (oracle/stark.c): the three sub-proofs self-verify (the reference's only pin for this
path, src/verifier_circuit.rs:253-254), public inputs equal the native SIPP chain, tampering is rejected."""
import numpy as np
import pytest

from oracle.py import sipp_native as sn
from tests import _oracle


@pytest.fixture(scope="module")
def case():
    A, B = sn.synthetic_inputs(2, 11)
    proof = sn.sipp_prove_native(A, B)
    ok, st, obl = sn.sipp_verify_native(A, B, proof, check_final_pairing=False)
    ios = sn.io_records(obl)
    return ios


@pytest.fixture(scope="module")
def g1_proof(case):
    return _oracle.stark_prove(0, case[0])


def test_g1_proof_verifies_and_binds_public_inputs(case, g1_proof):
    pf = g1_proof
    assert _oracle.stark_verify(pf) == 0
    n_pi = case[0].size
    assert (pf[-n_pi:] == case[0].reshape(-1)).all()
    assert pf[12] == len(pf)


def test_deterministic(case, g1_proof):
    assert (_oracle.stark_prove(0, case[0]) == g1_proof).all()


@pytest.mark.parametrize("where", ["cap", "opening", "final_poly", "pow", "query_row", "public_input"])
def test_tampered_proof_rejected(g1_proof, where):
    pf = g1_proof.copy()
    W, P = int(pf[4]), int(pf[5])
    caps = 16 + 3 * 64
    n_open = 2 * (2 * W + 2 * P + 4)
    rounds, flen = int(pf[8]), int(pf[9])
    fri = caps + n_open
    pos = {"cap": 16 + 5, "opening": caps + 7, "final_poly": fri + rounds * 64 + 1,
           "pow": fri + rounds * 64 + 2 * flen, "query_row": fri + rounds * 64 + 2 * flen + 1 + 3,
           "public_input": len(pf) - 3}[where]
    pf[pos] ^= 1
    assert _oracle.stark_verify(pf) != 0


def test_g2_and_fq12_verify(case):
    for kind in (1, 2):
        pf = _oracle.stark_prove(kind, case[kind])
        assert _oracle.stark_verify(pf) == 0


def test_proof_digests_match_committed_self_golden():
    """tests/golden/proof_digests_n4.json (tools/gen_golden.py digests): regenerate deliberately when the AIR changes."""
    import hashlib
    import json
    gold = json.load(open("tests/golden/proof_digests_n4.json"))
    d = np.load("tests/golden/sipp_n4_ios.npz")
    for kind, key in ((0, "g1"), (1, "g2"), (2, "fq12")):
        pf = _oracle.stark_prove(kind, d[key])
        assert len(pf) == gold[key]["words"] and int(pf[4]) == gold[key]["W"]
        assert hashlib.sha256(pf.tobytes()).hexdigest() == gold[key]["sha256"], key
