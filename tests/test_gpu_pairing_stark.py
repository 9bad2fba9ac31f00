"""The final-pairing STARK (API kind 6; reference src/bin/bls_aggregation.rs:76-77: `pairing_circuit(final_A, final_B)` connected to
`final_Z`) on the GPU against the CPU oracle: the value, the trace cell for cell, the proof word for word for 1 and 3 records
(the generator pairing among them), both verifiers' acceptance, and the refusals (wrong Z, points off the curve, Q outside G2)."""
import random

import numpy as np
import pytest

from tests import _oracle, _verify
from oracle.py import bn254 as bn

pytestmark = pytest.mark.gpu

SIPP_E_WITNESS = -8


def record(Pt, Q, Z=None):
    z = bn.pairing(Pt, Q) if Z is None else Z
    return np.array(bn.g1_to_u32(Pt) + bn.g2_to_u32(Q) + bn.f12_to_u32(z), dtype=np.uint32)


def points(seed):
    rnd = random.Random(seed)
    return bn.g1_mul(bn.G1, rnd.randrange(1, bn.R)), bn.g2_mul(bn.G2, rnd.randrange(1, bn.R))


@pytest.fixture(scope="module")
def ctx():
    import sipp_amd
    c = sipp_amd.Ctx(workspace_bytes=sipp_amd.lib().sipp_workspace_bytes(6, 16))
    yield c
    c.close()


@pytest.fixture(scope="module")
def recs3():
    return np.stack([record(bn.G1, bn.G2), record(*points(21)), record(*points(22))])


def test_outputs_are_the_oracles_pairings(ctx, recs3):
    """sipp_exp_outputs(kind 6): Z computed on the device and written into the records = oracle/py/bn254.py::pairing (arkworks' value)"""
    blank = recs3.copy()
    blank[:, 48:] = 0
    got = ctx.exp_outputs(6, blank)
    assert (got == recs3).all()
    assert (_oracle.pairing(recs3[1, :48]) == recs3[1, 48:]).all()


def test_edge_points_and_bilinearity(ctx):
    """the value kernel on the points the BLS example meets (-G1, small multiples, -G2 = [r - 1] G2) and on random ones, against the C
    reading; bilinearity e([2] P, [3] Q) = e(P, Q)^6 computed on the device on both sides"""
    rnd = random.Random(77)
    ps = [bn.G1, bn.g1_neg(bn.G1), bn.g1_mul(bn.G1, 2), bn.g1_mul(bn.G1, 3), bn.g1_mul(bn.G1, bn.R - 2)] + [bn.g1_mul(bn.G1, rnd.randrange(1, bn.R)) for _ in range(5)]
    qs = [bn.G2, bn.g2_mul(bn.G2, 2), bn.g2_neg(bn.G2), bn.g2_mul(bn.G2, 5), bn.g2_mul(bn.G2, bn.R - 3)] + [bn.g2_mul(bn.G2, rnd.randrange(1, bn.R)) for _ in range(5)]
    recs = np.zeros((len(ps), 144), dtype=np.uint32)
    for i, (a, b) in enumerate(zip(ps, qs)):
        recs[i, :48] = bn.g1_to_u32(a) + bn.g2_to_u32(b)
    got = ctx.exp_outputs(6, recs)
    for i in range(len(ps)):
        assert (got[i, 48:] == _oracle.pairing(recs[i, :48])).all(), i
    e = lambda i: [bn.u32_to_fq(list(got[i, 48 + 8 * k: 56 + 8 * k])) for k in range(12)]
    assert e(1) == bn.f12_inv(bn.f12_pow(e(0), 2))           # e(-G1, [2] G2) = e(G1, G2)^-2
    two_three = np.zeros((1, 144), dtype=np.uint32)
    two_three[0, :48] = bn.g1_to_u32(bn.g1_mul(ps[5], 2)) + bn.g2_to_u32(bn.g2_mul(qs[5], 3))
    lhs = ctx.exp_outputs(6, two_three)[0, 48:]
    assert [bn.u32_to_fq(list(lhs[8 * k: 8 * k + 8])) for k in range(12)] == bn.f12_pow(e(5), 6)
    # and all ten records in ONE proof (sixteen blocks of 2^13 rows, six of them padding)
    proof = ctx.prove(6, got)
    assert int(proof[2]) == 17 and int(proof[3]) == 16 and _verify.both_accept(proof)


def test_trace_matches_oracle_cell_for_cell(ctx, recs3):
    from sipp_amd._lib import to_host
    for recs in (recs3[1:2], recs3):
        ref = _oracle.Trace(6, recs)
        assert ref.air.table_bits == 8 and ref.air.log_rows == 13 and ref.log_n == (14 if len(recs) == 1 else 15)
        assert ctx.shape(6, len(recs))[:2] == (ref.log_n, ref.width)
        got = to_host(ctx.trace_build(6, recs))
        want = ref.array()
        assert got.shape == want.shape
        if not (got == want).all():
            bad = np.argwhere(got != want)
            raise AssertionError("%d cells differ; first (col,row): %s" % (len(bad), bad[:8].tolist()))


@pytest.mark.parametrize("count", [1, 3])
def test_proof_matches_oracle_word_for_word_and_verifies(ctx, recs3, count):
    recs = recs3[:count]
    proof = ctx.prove(6, recs)
    ref = _oracle.stark_prove(6, recs)
    assert len(proof) == len(ref)
    if not (proof == ref).all():
        raise AssertionError("first differing word: %d of %d" % (int(np.argmax(proof != ref)), len(ref)))
    assert _verify.both_accept(proof)
    assert int(proof[1]) == 6 and int(proof[3]) == (2 if count == 1 else 4)


def test_python_verifier_accepts_a_gpu_proof(recs3):
    import sipp_amd
    from oracle.py import stark_verify as sv
    cfg = sipp_amd.default_config()
    cfg.num_queries = 2
    cfg.pow_bits = 6
    c = sipp_amd.Ctx(workspace_bytes=sipp_amd.lib().sipp_workspace_bytes(6, 2), cfg=cfg)
    try:
        proof = c.prove(6, recs3[:1])
    finally:
        c.close()
    assert sv.verify(proof, dict(num_queries=2, pow_bits=6)) is None


def test_wrong_result_and_bad_points_are_refused_on_both_sides(ctx, recs3):
    import sipp_amd
    wrong = recs3[1:2].copy()
    wrong[0, 48] ^= 1                                   # another final_Z
    with pytest.raises(sipp_amd.SippError) as e:
        ctx.prove(6, wrong)
    assert e.value.code == SIPP_E_WITNESS
    with pytest.raises(RuntimeError):
        _oracle.stark_prove(6, wrong)
    Pt, Q = points(23)
    off = record((Pt[0], (Pt[1] + 1) % bn.P), Q, bn.F12_ONE).reshape(1, 144)       # P off the curve
    with pytest.raises(sipp_amd.SippError) as e:
        ctx.prove(6, off)
    assert e.value.code == SIPP_E_WITNESS
    big = recs3[1:2].copy()
    big[0, 0:8] = np.array(bn.fq_to_u32(bn.P), dtype=np.uint32)                  # a non-canonical word
    with pytest.raises(sipp_amd.SippError):
        ctx.prove(6, big)
    # the ctx still proves afterwards
    assert _verify.both_accept(ctx.prove(6, recs3[:1]))
