"""sipp_inner_product(s) -- the pairing products of the native SIPP prover (reference src/prover_native.rs:15-23), SURVEY.md
section 8(f) rank 3 -- against the CPU restatement oracle/py/bn254.py (optimal ate Miller loop + plain (p^12 - 1)/r power)."""
import numpy as np
import pytest

from oracle.py import bn254 as bn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import sipp_amd
    c = sipp_amd.Ctx(workspace_bytes=1 << 30)
    yield c
    c.close()


def points(rng, n):
    def scalar():
        return int.from_bytes(rng.bytes(32), "little") % bn.R or 1
    A = [bn.g1_mul(bn.G1, scalar()) for _ in range(n)]
    B = [bn.g2_mul(bn.G2, scalar()) for _ in range(n)]
    return A, B


def limbs(A, B):
    return (np.array([bn.g1_to_u32(a) for a in A], dtype=np.uint32), np.array([bn.g2_to_u32(b) for b in B], dtype=np.uint32))


@pytest.mark.parametrize("n", [1, 2, 5])
def test_inner_product_matches_the_oracle(ctx, n):
    rng = np.random.default_rng(100 + n)
    A, B = points(rng, n)
    g1, g2 = limbs(A, B)
    got = ctx.inner_products(g1, g2)[0]
    want = np.array(bn.f12_to_u32(bn.multi_pairing(A, B)), dtype=np.uint32)
    assert (got == want).all()


def test_generator_pairing_and_bilinearity(ctx):
    g1, g2 = limbs([bn.G1], [bn.G2])
    e = ctx.inner_products(g1, g2)[0]
    assert (e == np.array(bn.f12_to_u32(bn.pairing(bn.G1, bn.G2)), dtype=np.uint32)).all()
    # e([a] G1, [b] G2) == e([ab] G1, G2): two products of one pair each in ONE call
    a, b = 0x1234567890ABCDEF1234567, 0xFEDCBA09876543211234
    g1, g2 = limbs([bn.g1_mul(bn.G1, a), bn.g1_mul(bn.G1, a * b % bn.R)], [bn.g2_mul(bn.G2, b), bn.G2])
    both = ctx.inner_products(g1, g2, count=2)
    assert (both[0] == both[1]).all() and not (both[0] == e).all()


def test_round_shaped_call_and_infinity(ctx):
    """Z_L and Z_R of one SIPP round (src/prover_native.rs:51-52) in one call; a pair with the point at infinity contributes 1"""
    rng = np.random.default_rng(7)
    A, B = points(rng, 4)
    g1, g2 = limbs(A[2:] + A[:2], B[:2] + B[2:])           # (A2, B1) then (A1, B2)
    z = ctx.inner_products(g1, g2, count=2)
    assert (z[0] == np.array(bn.f12_to_u32(bn.multi_pairing(A[2:], B[:2])), dtype=np.uint32)).all()
    assert (z[1] == np.array(bn.f12_to_u32(bn.multi_pairing(A[:2], B[2:])), dtype=np.uint32)).all()
    g1, g2 = limbs(A[:2], B[:2])
    g1[1] = 0
    one_pair = ctx.inner_products(g1, g2)[0]
    assert (one_pair == np.array(bn.f12_to_u32(bn.pairing(A[0], B[0])), dtype=np.uint32)).all()


@pytest.mark.parametrize("n", [4, 8, 128, 256])
def test_native_chain_reproduces_the_fixtures(n):
    """sipp_prove_native + sipp_verify_native (reference src/prover_native.rs:26-80, src/verifier_native.rs:14-85) on the GPU:
    from A, B alone they reproduce the committed fixtures -- SIPPStatement limbs and every IO record of the three obligation
    lists -- which the CPU restatement oracle/py/sipp_native.py produced (tools/gen_golden.py), and the verifier accepts."""
    import sipp_amd
    d = np.load("tests/golden/sipp_n%d_ios.npz" % n)
    st = d["statement"]
    A, B = st[: 16 * n].reshape(n, 16), st[16 * n: 48 * n].reshape(n, 32)
    ctx = sipp_amd.Ctx(workspace_bytes=max(1 << 30, sipp_amd.lib().sipp_workspace_bytes(1, max(1, n // 2))))
    try:
        proof = ctx.prove_native(A, B)
        assert proof.shape == (2 * (n.bit_length() - 1) + 1, 96)
        ok, st2, ios = ctx.verify_native(A, B, proof)
        assert ok
        assert (st2 == st).all()
        for got, key in zip(ios, ("g1", "g2", "fq12")):
            assert got.shape == d[key].shape and (got == d[key]).all(), key
        # a tampered message is rejected (the folds no longer end in pairing(final_A, final_B))
        bad = proof.copy()
        bad[1, 0] ^= 1
        assert not ctx.verify_native(A, B, bad)[0]
    finally:
        ctx.close()


def test_whole_pipeline_from_points_to_three_proofs():
    """The flow of the reference's test_sipp_circuit (src/verifier_circuit.rs:192-269) with every heavy step in the product:
    A, B -> sipp_prove_native -> sipp_verify_native (statement + obligation lists) -> sipp_instance_prove -> three STARK
    proofs whose public inputs are exactly those obligations; each proof is checked by the oracle's verifier."""
    import sipp_amd
    from tests import _oracle, _verify
    n = 8
    d = np.load("tests/golden/sipp_n%d_ios.npz" % n)
    A, B = d["statement"][: 16 * n].reshape(n, 16), d["statement"][16 * n: 48 * n].reshape(n, 32)
    ctx = sipp_amd.Ctx(workspace_bytes=1 << 30)
    try:
        ok, st, ios = ctx.verify_native(A, B, ctx.prove_native(A, B))
        assert ok and (st == d["statement"]).all()
    finally:
        ctx.close()
    inst = sipp_amd.Instance([a.shape[0] for a in ios])
    try:
        proofs = inst.prove(ios)
        for kind in range(3):
            pf = proofs[kind]
            assert _verify.both_accept(pf)
            nio = int(pf[3])
            pis = pf[-nio * ios[kind].shape[1]:].reshape(nio, ios[kind].shape[1])
            assert (pis[: ios[kind].shape[0]] == ios[kind]).all()
    finally:
        inst.close()
    # the tail of the reference's test (src/verifier_circuit.rs:253-268): `data.prove(pw)` -> `data.verify(proof)` -> the proof's public
    # inputs read back as a SIPPStatement equal the native one.  The outer circuit here is the recursion-shaped STAND-IN (tools/plonk_synth.py:
    # it does not verify the three STARK proofs -- the reference's circuit lives in un-vendored crates), but the flow is the reference's:
    # public inputs = SIPPStatementTarget::to_vec() limbs (statements.rs:24-39), their hash bound by the PublicInput gate, witness
    # generation and prove() on the device, both verifiers of the outer proof, statement round trip.
    from tests.test_oracle_plonk import fri, _synth
    from tests.test_gpu_fri_generic import to_params
    from sipp_amd._lib import to_device
    log_n, pis = 10, [int(x) for x in st]
    assert len(pis) == 48 * n + 240
    ps, circ, wires, cs, gate, _pis, pih = _synth(log_n, 136, 80, seed=3, pis=pis)
    K = circ["num_constants"]
    ctx = sipp_amd.Ctx(workspace_bytes=2 << 30)
    try:
        d_w, d_cs = to_device(ps.blank_generated(circ, wires, gate)), to_device(cs)
        ctx.plonk_generate_witness(d_w, d_cs[:K], log_n, ps.generators(circ), pih)
        ofp = fri(log_n, rate_bits=3, cap_height=4, nq=28, arity=4, fpb=5)
        op, digest = _oracle.plonk_params(80, 8, 2), (0x53495050, 1, 2, 3)
        outer = ctx.plonk_prove_gates(d_w, d_cs, log_n, sipp_amd.PlonkParams(80, 8, 2), to_params(ofp), sipp_amd.PlonkCircuit.from_dict(circ), digest, pis)
    finally:
        ctx.close()
    cs_cap = _oracle.Batch(cs, log_n, rate_bits=3, cap_height=4).cap
    assert _verify.lib_plonk_verify(outer, cs_cap, op, ofp, circ, digest) == 0 and _oracle.plonk_verify_gates(outer, cs_cap, op, ofp, circ, digest) == 0
    assert (outer[-len(pis):].astype(np.uint32) == d["statement"]).all()              # SIPPStatement::from_vec(proof.public_inputs) == statement
    forged = outer.copy()
    forged[-1] ^= 1                                                                     # another final_Z limb: the hash in the transcript no longer matches
    assert _verify.lib_plonk_verify(forged, cs_cap, op, ofp, circ, digest) != 0 and _oracle.plonk_verify_gates(forged, cs_cap, op, ofp, circ, digest) != 0


@pytest.mark.parametrize("n", [300, 513])
def test_large_products_by_bilinearity(ctx, n):
    """more pairs than one workgroup has lanes (strided partial products + tree): prod_i e([a_i] G1, [b_i] G2) must equal
    e([sum a_i b_i] G1, G2) -- a size-independent property, both sides computed on the device"""
    rng = np.random.default_rng(n)
    a = [int.from_bytes(rng.bytes(6), "little") + 1 for _ in range(n)]       # 48-bit scalars: the host side (big-int point multiples) is the
    b = [int.from_bytes(rng.bytes(6), "little") + 1 for _ in range(n)]       # test's time, the property does not depend on their size
    g1, g2 = limbs([bn.g1_mul(bn.G1, x) for x in a], [bn.g2_mul(bn.G2, y) for y in b])
    lhs = ctx.inner_products(g1, g2)[0]
    s = sum(x * y for x, y in zip(a, b)) % bn.R
    h1, h2 = limbs([bn.g1_mul(bn.G1, s)], [bn.G2])
    rhs = ctx.inner_products(h1, h2)[0]
    assert (lhs == rhs).all()


def test_native_chain_edge_sizes_and_errors(ctx):
    """n = 1 (no round: the proof is Z alone) and n = 2 against the CPU restatement of the chain; sizes that are not a power
    of two are refused with SIPP_E_BADARG"""
    import sipp_amd
    from oracle.py import sipp_native as sn
    rng = np.random.default_rng(21)
    for n in (1, 2):
        A, B = points(rng, n)
        g1, g2 = limbs(A, B)
        proof = ctx.prove_native(g1, g2)
        want = sn.sipp_prove_native(A, B)
        assert proof.shape[0] == len(want) == 2 * (n.bit_length() - 1) + 1
        for got, w in zip(proof, want):
            assert (got == np.array(bn.f12_to_u32(w), dtype=np.uint32)).all()
        ok, st, ios = ctx.verify_native(g1, g2, proof)
        ok2, st2, obl = sn.sipp_verify_native(A, B, want)
        assert ok and ok2
        assert (st == np.array(sn.statement_to_u32(st2), dtype=np.uint32)).all()
        if n > 1:
            for got, w in zip(ios, sn.io_records(obl)):
                assert (got == w).all()
    A, B = points(rng, 3)
    g1, g2 = limbs(A, B)
    with pytest.raises(sipp_amd.SippError) as e:
        ctx.prove_native(g1, g2)
    assert e.value.code == -1
