"""Native SIPP restatement (reference src/prover_native.rs, src/verifier_native.rs, src/transcript_native.rs):
what the reference's own test_sipp_native asserts (src/verifier_native.rs:96-106), on seeded inputs."""
import numpy as np
import pytest

from oracle.py import bn254 as bn
from oracle.py import sipp_native as sn


@pytest.fixture(scope="module")
def sipp8():
    A, B = sn.synthetic_inputs(8, 0x51515050)
    proof = sn.sipp_prove_native(A, B)
    ok, st, obl = sn.sipp_verify_native(A, B, proof)
    return A, B, proof, ok, st, obl


def test_sipp_native_roundtrip(sipp8):
    A, B, proof, ok, st, obl = sipp8
    assert ok                                            # e(final_A, final_B) == final_Z
    assert len(proof) == 2 * 3 + 1                       # src/verifier_circuit.rs:219
    assert sn.inner_product(A, B) == proof[-1]           # src/verifier_native.rs:105
    assert len(obl["g1"]) == 7 and len(obl["g2"]) == 7 and len(obl["fq12"]) == 6
    assert len(sn.statement_to_u32(st)) == 16 * 8 + 32 * 8 + 96 + 16 + 32 + 96   # src/statements.rs:41-45


def test_bad_proof_rejected(sipp8):
    A, B, proof, *_ = sipp8
    bad = list(proof)
    bad[0] = bn.f12_mul(bad[0], bad[0])
    ok, _, _ = sn.sipp_verify_native(A, B, bad)
    assert not ok


def test_transcript_challenge_quirk():
    t = sn.Transcript()
    t.append([1, 2, 3])
    c = t.get_challenge()
    assert 0 < c < bn.R
    # limbs: 4 hash words -> (lo, hi) u32 digits, little endian (src/transcript_native.rs:56-65)
    d = sn.hash_no_pad(t.state)
    assert c == sum(x << (64 * i) for i, x in enumerate(d)) % bn.R   # holds whenever every word >= 2^32


def test_io_records_shape(sipp8):
    *_, obl = sipp8
    g1, g2, f12 = sn.io_records(obl)
    assert g1.shape == (7, 56) and g2.shape == (7, 104) and f12.shape == (6, 296)
