"""The hardened G1 / G2 exponentiation AIRs (API kinds 4 / 5) on the GPU against the oracle: trace cell for cell, proof word for word
(u8 variant, n = 4), the u16 variant at the n = 128 size through the oracle's verifier, and the crafted record of
tests/test_oracle_hardened.py refused."""
import numpy as np
import pytest

from tests import _oracle

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
    assert _oracle.stark_verify(pf) == 0
    # the plain kind on the same ctx afterwards: its own program, its own proof (tables are cached per AIR variant)
    assert (ctx.prove(kind - 4, ios) == _oracle.stark_prove(kind - 4, ios)).all()


@pytest.mark.parametrize("kind", [4, 5])
def test_u16_variant_at_the_n128_size_verifies(kind):
    import sipp_amd
    d = np.load("tests/golden/sipp_n128_ios.npz")
    ios = d[("g1", "g2")[kind - 4]]
    c = sipp_amd.Ctx(workspace_bytes=sipp_amd.lib().sipp_workspace_bytes(kind, ios.shape[0]))
    try:
        assert c.shape(kind, ios.shape[0]) == ((16, 723, 410, 4), (16, 1436, 820, 4))[kind - 4]
        pf = c.prove(kind, ios)
    finally:
        c.close()
    assert _oracle.stark_verify(pf) == 0
    nio = int(pf[3])
    assert (pf[-nio * ios.shape[1]:].reshape(nio, ios.shape[1])[: ios.shape[0]] == ios).all()


def test_the_crafted_record_is_refused(ctx):
    import sipp_amd
    from oracle.py import bn254 as bn
    from oracle.py import sipp_native as sn
    x = bn.g1_mul(bn.G1, 77)
    rec = np.array([bn.g1_to_u32(x) + bn.g1_to_u32(x) + sn.exp_to_u32(3) + bn.g1_to_u32(bn.g1_mul(x, 4))] * 2, dtype=np.uint32)
    for kind in (0, 4):
        with pytest.raises(sipp_amd.SippError) as e:
            ctx.prove(kind, rec)
        assert e.value.code == -8
