"""MapToG2 (kind 3; SURVEY.md section 8f rank 4, reference src/bin/bls_aggregation.rs:65, :100-104) on the GPU against the two
CPU readings: the map's values, the trace cell for cell, the proof word for word, the cofactor clearing as G2ExpStark obligations."""
import os
import sys

import numpy as np
import pytest

from tests import _oracle, _verify

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle", "py"))

pytestmark = pytest.mark.gpu


def messages(n, seed=11):
    import bn254
    import random
    rnd = random.Random(seed)
    us = [(rnd.randrange(bn254.P), rnd.randrange(bn254.P)) for _ in range(n)]
    us[0] = (0, 5)           # sgn0 decided by the second coordinate
    if n > 2:
        us[1] = (7, 0)
        us[2] = (0, 0)       # u = 0: tv1 = 0, x1 = x2 = -Z/2 +- 0
    if n > 4:                # u^2 g(Z) = 1: the inverted product is zero (inv0)
        import map_to_g2 as M
        r = M.sqrt_even(bn254.f2_inv(M.C1)) or M.sqrt_even(bn254.f2_neg(bn254.f2_inv(M.C1)))
        us[3], us[4] = r, bn254.f2_neg(r)
    return us, np.array([bn254.fq_to_u32(u[0]) + bn254.fq_to_u32(u[1]) for u in us], dtype=np.uint32)


@pytest.fixture(scope="module")
def ctx():
    import sipp_amd
    c = sipp_amd.Ctx(workspace_bytes=max(sipp_amd.lib().sipp_workspace_bytes(3, 40), sipp_amd.lib().sipp_workspace_bytes(1, 80)))
    yield c
    c.close()


def test_map_and_cofactor_clearing_match_the_python_reading(ctx):
    import bn254
    import map_to_g2 as M
    us, words = messages(20)
    recs, g2, pts = ctx.map_to_g2(words)
    assert (recs == _oracle.map_to_g2(words)).all()          # C reading
    branches = set()
    for i, u in enumerate(us):
        w = M.witness(u)
        branches.add((w["e1"], w["e2"]))
        q = (w["XS"], w["Y"])
        assert list(recs[i, 16:]) == bn254.g2_to_u32(q)
        cleared = bn254.g2_mul(q, bn254.G2_COFACTOR)
        assert list(pts[i]) == bn254.g2_to_u32(cleared)
        assert bn254.g2_mul(cleared, bn254.R) is None        # in the r-torsion
        # the two obligations per message: G + [h] Q, then - G
        assert list(g2[i, :32]) == bn254.g2_to_u32(q) and list(g2[i, 32:64]) == bn254.g2_to_u32(bn254.G2)
        assert list(g2[20 + i, 72:]) == bn254.g2_to_u32(cleared) and list(g2[20 + i, 32:64]) == list(g2[i, 72:])
    assert len(branches) == 3, branches                      # x1, x2 and x3 all taken
    # the obligations are ordinary G2ExpStark records: the oracle's generator accepts them (claimed outputs checked)
    t = _oracle.Trace(1, g2[[0, 1, 20, 21]])
    assert t.check_row(0) == -1 and t.check_row(1023) == -1


def test_trace_matches_oracle_cell_for_cell(ctx):
    from sipp_amd._lib import to_host
    _, words = messages(37)
    recs = ctx.map_to_g2(words, cofactor=False)
    ref = _oracle.Trace(3, recs)
    assert ref.log_n == 10 and ref.air.table_bits == 8 and ref.air.log_rows == 3 and ref.num_io == 128
    assert ctx.shape(3, 37)[:2] == (10, ref.width)
    got = to_host(ctx.trace_build(3, recs))
    want = ref.array()
    assert got.shape == want.shape
    if not (got == want).all():
        bad = np.argwhere(got != want)
        raise AssertionError("%d cells differ; first (col,row): %s" % (len(bad), bad[:8].tolist()))


def test_proof_matches_oracle_word_for_word_and_verifies(ctx):
    _, words = messages(5, seed=3)
    recs = ctx.map_to_g2(words, cofactor=False)
    proof = ctx.prove(3, recs)
    assert _verify.both_accept(proof)
    want = _oracle.stark_prove(3, recs)
    assert proof.shape == want.shape and (proof == want).all()


def test_wrong_or_unprovable_records_are_refused(ctx):
    import sipp_amd
    import bn254
    _, words = messages(4, seed=5)
    recs = ctx.map_to_g2(words, cofactor=False)
    bad = recs.copy()
    # the other root: on the curve, wrong sign
    y = (bn254.u32_to_fq(list(bad[1, 32:40])), bn254.u32_to_fq(list(bad[1, 40:48])))
    bad[1, 32:] = bn254.fq_to_u32((-y[0]) % bn254.P) + bn254.fq_to_u32((-y[1]) % bn254.P)
    with pytest.raises(sipp_amd.SippError) as e:
        ctx.prove(3, bad)
    assert e.value.code == -8                     # SIPP_E_WITNESS
    bad = recs.copy()
    bad[2, 0:8] = bn254.fq_to_u32(bn254.P - 1)
    bad[2, 7] |= 0x80000000                      # u.c0 >= p
    with pytest.raises(sipp_amd.SippError) as e:
        ctx.prove(3, bad)
    assert e.value.code == -8
    # bad arguments of the native entry point: status, no crash
    L = sipp_amd.lib()
    out = np.zeros((4, 48), dtype=np.uint32)
    assert L.sipp_map_to_g2(ctx.h, None, 4, out.ctypes.data, None, None) == -1            # SIPP_E_BADARG
    assert L.sipp_map_to_g2(ctx.h, words.ctypes.data, 0, out.ctypes.data, None, None) == -1
    assert L.sipp_map_to_g2(ctx.h, words.ctypes.data, 4, None, None, None) == -1
    big = np.zeros(((1 << 17) + 1, 48), dtype=np.uint32)                                   # more records than a STARK of this kind takes
    assert L.sipp_map_to_g2(ctx.h, np.zeros(((1 << 17) + 1, 16), dtype=np.uint32).ctypes.data, (1 << 17) + 1, big.ctypes.data, None, None) == -1
    # the ctx survives
    assert _verify.both_accept(ctx.prove(3, recs))


def test_u16_variant_at_2_to_the_16_rows_is_accepted_by_the_oracle_verifier():
    """more than 4096 messages: 2^16 rows and more, where the range table is the u16 one (one checked cell per limb: W = 1720,
    P = 756).  The oracle's verifier replays the proof."""
    import sipp_amd
    _, words = messages(40, seed=9)
    L = sipp_amd.lib()
    n = 5000
    c = sipp_amd.Ctx(workspace_bytes=L.sipp_workspace_bytes(3, n))
    try:
        assert c.shape(3, n) == (16, 1720, 756, 4)
        recs40 = c.map_to_g2(words, cofactor=False)
        recs = recs40[np.arange(n) % 40]
        proof = c.prove(3, recs)
    finally:
        c.close()
    assert (int(proof[1]), int(proof[2]), int(proof[3])) == (3, 16, 8192)
    assert _verify.both_accept(proof)
    proof[16 + 5] ^= 1
    assert _verify.both_refuse(proof)


def test_two_thousand_random_messages_map_like_the_c_reading(ctx):
    """values only (no proof): every branch and both sign cases many times over; the C reading takes ~2 ms per message"""
    rng = np.random.default_rng(20261004)
    words = np.zeros((2000, 16), dtype=np.uint32)
    words[:, :8] = rng.integers(0, 2**32, size=(2000, 8), dtype=np.uint64)
    words[:, 8:] = rng.integers(0, 2**32, size=(2000, 8), dtype=np.uint64)
    words[:, 7] &= 0x1FFFFFFF          # < 2^253 < p
    words[:, 15] &= 0x1FFFFFFF
    words[::7, 8:] = 0                 # u in Fp
    words[::11, :8] = 0                # u = c u: sgn0 from the second coordinate
    got = ctx.map_to_g2(words, cofactor=False)
    want = _oracle.map_to_g2(words)
    assert (got == want).all(), np.argwhere((got != want).any(axis=1))[:8].tolist()


def test_golden_vectors(ctx):
    """tests/golden/mapg2_vectors.json: records, cleared points and the proof digest from the GPU"""
    import hashlib
    import json
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "mapg2_vectors.json")))
    want = np.array(g["records"], dtype=np.uint32)
    recs, _, pts = ctx.map_to_g2(want[:, :16])
    assert (recs == want).all() and (pts == np.array(g["cleared"], dtype=np.uint32)).all()
    assert hashlib.sha256(ctx.prove(3, recs).tobytes()).hexdigest() == g["proof"]["sha256"]


def test_gpu_proof_is_accepted_by_the_python_reading_of_the_verifier():
    """the chain GPU prover -> oracle/py/stark_verify.py, with no C oracle in between (a 3-query configuration: pure-Python hashing)"""
    import sipp_amd
    from oracle.py import stark_verify as sv
    cfg = sipp_amd.default_config()
    cfg.num_queries, cfg.pow_bits = 3, 6
    _, words = messages(6, seed=21)
    c = sipp_amd.Ctx(cfg=cfg, workspace_bytes=2 << 30)
    try:
        recs = c.map_to_g2(words, cofactor=False)
        pf = c.prove(3, recs)
    finally:
        c.close()
    assert sv.verify(pf, dict(num_queries=3, pow_bits=6)) is None
    bad = pf.copy()
    bad[16 + 3 * 64 + 2 * 9] = (int(bad[16 + 3 * 64 + 2 * 9]) + 1) % _oracle.P          # an opening at zeta
    assert sv.verify(bad, dict(num_queries=3, pow_bits=6)) == "quotient identity 0"
