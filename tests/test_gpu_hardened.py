"""The hardened G1 / G2 exponentiation AIRs (API kinds 4 / 5) on the GPU against the oracle: trace cell for cell, proof word for word
(u8 variant, n = 4), the u16 variant at the n = 128 size through the oracle's verifier, and the crafted records of
tests/test_oracle_hardened.py: refused by the plain kinds, proved by the hardened ones."""
import numpy as np
import pytest

from tests import _oracle, _verify

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ios4():
    d = np.load("tests/golden/sipp_n4_ios.npz")
    return d["g1"], d["g2"]


@pytest.fixture(scope="module")
def ctx():
    import sipp_amd
    c = sipp_amd.Ctx(workspace_bytes=4 << 30)
    yield c
    c.close()


@pytest.mark.parametrize("kind", [4, 5])
def test_trace_and_proof_match_the_oracle(ctx, ios4, kind):
    from sipp_amd._lib import to_host
    ios = ios4[kind - 4]
    ref = _oracle.Trace(kind, ios)
    assert ctx.shape(kind, ios.shape[0])[:2] == (ref.log_n, ref.width)
    got = to_host(ctx.trace_build(kind, ios))
    want = ref.array()
    if not (got == want).all():
        bad = np.argwhere(got != want)
        raise AssertionError("%d cells differ; first (col,row): %s" % (len(bad), bad[:8].tolist()))
    pf = ctx.prove(kind, ios)
    want = _oracle.stark_prove(kind, ios)
    assert int(pf[1]) == kind and pf.shape == want.shape and (pf == want).all()
    assert _verify.both_accept(pf)
    # the plain kind on the same ctx afterwards: its own program, its own proof (tables are cached per AIR variant)
    assert (ctx.prove(kind - 4, ios) == _oracle.stark_prove(kind - 4, ios)).all()


@pytest.mark.parametrize("kind", [4, 5])
def test_u16_variant_at_the_n128_size_verifies(kind):
    import sipp_amd
    d = np.load("tests/golden/sipp_n128_ios.npz")
    ios = d[("g1", "g2")[kind - 4]]
    c = sipp_amd.Ctx(workspace_bytes=sipp_amd.lib().sipp_workspace_bytes(kind, ios.shape[0]))
    try:
        assert c.shape(kind, ios.shape[0]) == ((16, 762, 410, 4), (16, 1490, 820, 4))[kind - 4]
        pf = c.prove(kind, ios)
    finally:
        c.close()
    assert _verify.both_accept(pf)
    nio = int(pf[3])
    assert (pf[-nio * ios.shape[1]:].reshape(nio, ios.shape[1])[: ios.shape[0]] == ios).all()


def test_records_that_meet_the_running_power_plain_refuses_hardened_proves(ctx):
    """offset = [2^i - (e mod 2^i)] x with bit i of e set (the accumulator equals the running power on add row i), G1 and G2: the plain
    kinds return SIPP_E_WITNESS, the hardened kinds prove them -- trace and proof equal to the oracle's; likewise where the accumulator
    meets MINUS the running power (the sum is the identity); only an OUTPUT at the identity is refused by both"""
    import sipp_amd
    from sipp_amd._lib import to_host
    from oracle.py import bn254 as bn
    from oracle.py import sipp_native as sn
    cases = ((3, 0), (0b1101, 2), ((1 << 40) | 5, 40))
    x1, x2 = bn.g1_mul(bn.G1, 1234567), bn.g2_mul(bn.G2, 7654321)
    g1, g2 = [], []
    for e, i in cases:
        k = (1 << i) - (e % (1 << i))
        g1.append(bn.g1_to_u32(x1) + bn.g1_to_u32(bn.g1_mul(x1, k)) + sn.exp_to_u32(e) + bn.g1_to_u32(bn.g1_mul(x1, (k + e) % bn.R)))
        g2.append(bn.g2_to_u32(x2) + bn.g2_to_u32(bn.g2_mul(x2, k)) + sn.exp_to_u32(e) + bn.g2_to_u32(bn.g2_mul(x2, (k + e) % bn.R)))
    for base, recs in ((0, np.array(g1, dtype=np.uint32)), (1, np.array(g2, dtype=np.uint32))):
        with pytest.raises(sipp_amd.SippError) as e:
            ctx.prove(base, recs)
        assert e.value.code == -8
        # the outputs alone come from the complete scan: right for these records in either kind
        assert (ctx.exp_outputs(base + 4, recs) == recs).all()
        ref = _oracle.Trace(base + 4, recs)
        got = to_host(ctx.trace_build(base + 4, recs))
        want = ref.array()
        if not (got == want).all():
            bad = np.argwhere(got != want)
            raise AssertionError("kind %d: %d cells differ; first (col,row): %s" % (base + 4, len(bad), bad[:8].tolist()))
        pf = ctx.prove(base + 4, recs)
        assert (pf == _oracle.stark_prove(base + 4, recs)).all() and _verify.both_accept(pf)
    # R = -P on a used addition: the accumulator passes through the identity (offset = -x, odd exponent; and a later meeting:
    # offset = -[2^i + (e mod 2^i)] x with bit i set)
    g1n, g2n = [], []
    for e, i in ((3, 0), (0b1001, 0), (0b10110, 2), ((1 << 200) | 1, 0)):
        k = (1 << i) + (e % (1 << i))
        g1n.append(bn.g1_to_u32(x1) + bn.g1_to_u32(bn.g1_neg(bn.g1_mul(x1, k))) + sn.exp_to_u32(e) + bn.g1_to_u32(bn.g1_mul(x1, (e - k) % bn.R)))
        g2n.append(bn.g2_to_u32(x2) + bn.g2_to_u32(bn.g2_neg(bn.g2_mul(x2, k))) + sn.exp_to_u32(e) + bn.g2_to_u32(bn.g2_mul(x2, (e - k) % bn.R)))
    for base, recs in ((0, np.array(g1n, dtype=np.uint32)), (1, np.array(g2n[:3], dtype=np.uint32))):
        with pytest.raises(sipp_amd.SippError) as e:
            ctx.prove(base, recs)
        assert e.value.code == -8
        assert (ctx.exp_outputs(base + 4, recs) == recs).all()
        got = to_host(ctx.trace_build(base + 4, recs))
        ref = _oracle.Trace(base + 4, recs)          # (kept alive: array() is a view of its buffer)
        want = ref.array()
        if not (got == want).all():
            bad = np.argwhere(got != want)
            raise AssertionError("kind %d: %d cells differ; first (col,row): %s" % (base + 4, len(bad), bad[:8].tolist()))
        pf = ctx.prove(base + 4, recs)
        assert (pf == _oracle.stark_prove(base + 4, recs)).all() and _verify.both_accept(pf)
    # the OUTPUT at the identity: no record can say it; refused by both variants
    none = np.array([bn.g1_to_u32(x1) + bn.g1_to_u32(bn.g1_neg(x1)) + sn.exp_to_u32(1) + bn.g1_to_u32(x1)] * 2, dtype=np.uint32)
    for kind in (0, 4):
        with pytest.raises(sipp_amd.SippError) as e:
            ctx.prove(kind, none)
        assert e.value.code == -8


def test_hardened_ctx_flag_through_the_instance_entry_point(ctx, ios4):
    """sipp_ctx_set_hardened: kinds 0 / 1 on such a ctx mean 4 / 5 in every entry point -- sipp_instance_prove gives the proofs
    sipp_prove(4 / 5) gives, Fq12 is untouched, and the flag can be taken back"""
    import sipp_amd
    d = np.load("tests/golden/sipp_n4_ios.npz")
    ios = [d[k] for k in ("g1", "g2", "fq12")]
    inst = sipp_amd.Instance([a.shape[0] for a in ios], hardened=True)
    try:
        proofs = [p.copy() for p in inst.prove(ios)]
        assert [int(p[1]) for p in proofs] == [4, 5, 2]
        assert inst.ctxs[0].shape(0, ios[0].shape[0]) == ctx.shape(4, ios[0].shape[0])
        c0 = inst.ctxs[0]
        c0._ck(c0.L.sipp_ctx_set_hardened(c0.h, 0), "set_hardened")
        plain = c0.prove(0, ios[0])
        assert int(plain[1]) == 0 and (plain == ctx.prove(0, ios[0])).all()
    finally:
        inst.close()
    for k in (0, 1):
        assert (proofs[k] == ctx.prove(k + 4, ios[k])).all()
    assert (proofs[2] == ctx.prove(2, ios[2])).all()
    assert all(_verify.both_accept(p) for p in proofs)


def test_hardened_queue(ctx):
    """sipp_instances_prove over slots whose ctxs carry the hardened flag: every instance's G1 / G2 proofs are the kind 4 / 5 proofs"""
    import sipp_amd
    d = np.load("tests/golden/sipp_n4_ios.npz")
    ios = [d[k] for k in ("g1", "g2", "fq12")]
    q = sipp_amd.InstanceQueue([a.shape[0] for a in ios], in_flight=2, hardened=True)
    try:
        res = q.prove([ios, ios, ios])
    finally:
        q.close()
    want = [ctx.prove(4, ios[0]), ctx.prove(5, ios[1]), ctx.prove(2, ios[2])]
    for inst in res:
        for k in range(3):
            assert (inst[k] == want[k]).all()
