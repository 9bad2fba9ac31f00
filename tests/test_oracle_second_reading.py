"""oracle/py/plonky2_generic.py (pure-Python, written independently of oracle/*.c) against the C oracle:
Poseidon / sponge vectors, FFT / coset LDE, PolynomialBatch caps, Merkle paths, and a complete replay of the
Fiat-Shamir transcript + FRI verification of C-oracle proofs (every challenge, the proof-of-work rule, all query rounds).
Two readings of upstream's published structure agreeing is not a pin to upstream (PARITY UNPINNED) -- it removes the
single-reading risk VERDICT round 1 named."""
import numpy as np
import pytest

from oracle.py import plonky2_generic as g2
from tests import _oracle
from tests.test_oracle_generic import KAT

P = g2.P


def test_poseidon_naive_form_reproduces_the_kats_and_the_c_oracle():
    for inp, out in KAT:
        assert g2.poseidon(inp) == out
    rng = np.random.default_rng(5)
    for _ in range(20):
        s = [int(x) for x in _oracle.rand_field(rng, 12)]
        assert g2.poseidon(s) == [int(x) for x in _oracle.permute(s)]
    for n in (1, 4, 5, 8, 9, 16, 23):
        v = [int(x) for x in _oracle.rand_field(rng, n)]
        assert g2.hash_no_pad(v) == [int(x) for x in _oracle.hash_no_pad(v)]
    a, b = [int(x) for x in _oracle.rand_field(rng, 4)], [int(x) for x in _oracle.rand_field(rng, 4)]
    assert g2.two_to_one(a, b) == [int(x) for x in _oracle.two_to_one(a, b)]
    assert g2.hash_or_noop([3, 4]) == [3, 4, 0, 0]


def test_fft_and_coset_lde_agree(oracle):
    rng = np.random.default_rng(6)
    for log_n in (1, 3, 7):
        c = _oracle.rand_field(rng, 1 << log_n)
        v = c.copy()
        oracle.orc_fft(v, log_n)
        assert g2.fft([int(x) for x in c]) == [int(x) for x in v]
        assert g2.ifft([int(x) for x in v]) == [int(x) for x in c]
        out = np.zeros(4 << log_n, dtype=np.uint64)
        oracle.orc_coset_lde(c, log_n, 2, 7, out)
        assert g2.coset_fft(g2.lde([int(x) for x in c], 2), 7) == [int(x) for x in out]


@pytest.mark.parametrize("ncols,log_n,cap_height", [(3, 4, 2), (11, 5, 4), (9, 3, 4), (20, 4, 0)])
def test_polynomial_batch_caps_and_paths_agree(ncols, log_n, cap_height):
    rng = np.random.default_rng(ncols)
    vals = _oracle.rand_field(rng, (ncols, 1 << log_n))
    cb = _oracle.Batch(vals, log_n, rate_bits=1, cap_height=cap_height)
    pb = g2.PolynomialBatch.from_values([[int(x) for x in col] for col in vals], 1, min(cap_height, log_n + 1))
    assert pb.polynomials == [[int(x) for x in col] for col in cb.coeffs]
    assert pb.tree.leaves == [[int(x) for x in row] for row in cb.leaves]
    assert pb.tree.cap == [[int(x) for x in d] for d in cb.cap]
    idx = 5
    sib = pb.tree.prove(idx)
    assert g2.verify_merkle_proof_to_cap(pb.tree.leaves[idx], idx, pb.tree.cap, sib)
    assert not g2.verify_merkle_proof_to_cap(pb.tree.leaves[idx], idx ^ 1, pb.tree.cap, sib)


def replay(pf, cfg):
    """Fiat-Shamir + FRI verification of a flat proof (layout: oracle/stark.c header) with the Python reading only.
    The constraint check at zeta is AIR-specific and stays with the C verifier."""
    pf = [int(x) for x in pf]
    kind, log_n, nio, W, Pz, Q, cap_h, rounds, flen, nq, ppi, total = pf[1:13]
    assert total == len(pf)
    pos = [16]

    def take(n):
        v = pf[pos[0]:pos[0] + n]
        pos[0] += n
        return v

    def take_cap():
        return [take(4) for _ in range(1 << cap_h)]

    pis = pf[len(pf) - nio * ppi:]
    ch = g2.Challenger()
    # this repository's statement binding (oracle/stark.c header), built from the generic primitives
    ch.observe_many([kind, log_n, nio, W, Pz, Q, cfg.rate_bits, cfg.cap_height, cfg.pow_bits, cfg.arity_bits,
                     cfg.final_poly_bits, cfg.num_queries, cfg.num_challenges, cfg.pow_rule, ppi, 0])
    ch.observe_many(g2.MerkleTree([pis[i * ppi:(i + 1) * ppi] for i in range(nio)], 0).cap[0])   # pi_root
    trace_cap = take_cap()
    ch.observe_cap(trace_cap)
    perm = [(ch.get(), ch.get()) for _ in range(cfg.num_challenges)]          # (beta, gamma) per challenge set
    z_cap = take_cap()
    ch.observe_cap(z_cap)
    alphas = ch.get_n(cfg.num_challenges)
    q_cap = take_cap()
    ch.observe_cap(q_cap)
    zeta = ch.get_ext()
    ext_list = lambda n: [g2.Ext(*take(2)) for _ in range(n)]
    local, nxt, z, znext, quot = ext_list(W), ext_list(W), ext_list(Pz), ext_list(Pz), ext_list(Q)
    for e in local + z + quot + nxt + znext:                                  # zeta batch, then g*zeta batch
        ch.observe_ext(e)
    alpha = ch.get_ext()
    arities = g2.reduction_arity_bits(cfg.arity_bits, cfg.final_poly_bits, log_n, cfg.rate_bits, cfg.cap_height)
    assert len(arities) == rounds
    caps, betas = [], []
    for _ in arities:
        caps.append(take_cap())
        ch.observe_cap(caps[-1])
        betas.append(ch.get_ext())
    final_poly = ext_list(flen)
    assert flen == (1 << log_n) >> sum(arities)
    for c in final_poly:
        ch.observe_ext(c)
    witness = take(1)[0]
    assert g2.pow_ok(g2.pow_response(ch, cfg.pow_rule, witness), cfg.pow_bits), "proof of work"
    log_m = log_n + cfg.rate_bits
    gz = zeta * g2.primitive_root_of_unity(log_n)
    for _ in range(nq):
        x = ch.get() % (1 << log_m)
        rows = []
        for ncols, cap in ((W, trace_cap), (Pz, z_cap), (Q, q_cap)):
            row, sib = take(ncols), [take(4) for _ in range(log_m - cap_h)]
            assert g2.verify_merkle_proof_to_cap(row, x, cap, sib), "initial tree path"
            rows.append(row)
        steps, xi = [], x
        for r, ab in enumerate(arities):
            ev = take(2 << ab)
            xi >>= ab
            ns = max(0, log_m - sum(arities[:r + 1]) - cap_h)
            sib = [take(4) for _ in range(ns)]
            assert g2.verify_merkle_proof_to_cap(ev, xi, caps[r], sib), "fri layer path"
            steps.append([g2.Ext(ev[2 * k], ev[2 * k + 1]) for k in range(1 << ab)])
        batches = [(zeta, rows[0] + rows[1] + rows[2], local + z + quot), (gz, rows[0] + rows[1], nxt + znext)]
        err = g2.fri_verify_query(x, log_n, cfg.rate_bits, arities, alpha, batches, rows, steps, betas, final_poly)
        assert err is None, err
    assert pos[0] + nio * ppi == len(pf)
    return dict(perm=perm, alphas=alphas, zeta=zeta, alpha=alpha, betas=betas, witness=witness)


@pytest.mark.parametrize("pow_rule", [0, 1])
def test_python_reading_verifies_the_c_oracles_proof(pow_rule):
    """G1ExpStark proof of the n = 4 fixture by oracle/stark.c, replayed by the Python reading: transcript order,
    challenge pop order, extension challenges, Merkle path convention, combine / fold / final-poly checks of every
    query, and the proof-of-work rule (both recollections) must all agree, or an assertion names the first difference."""
    ios = np.load("tests/golden/sipp_n4_ios.npz")["g1"]
    cfg = _oracle.default_config()
    cfg.pow_rule = pow_rule
    cfg.num_queries = 12
    cfg.pow_bits = 12
    pf = _oracle.stark_prove(0, ios, cfg)
    assert _oracle.stark_verify(pf, cfg) == 0
    out = replay(pf, cfg)
    # the grind rule: the C oracle's witness is the smallest valid one under the Python reading too
    assert out["witness"] < 1 << 20
    bad = pf.copy()
    bad[16 + 192 + 5] ^= 1                                   # an opened value: FRI consistency must break
    with pytest.raises(AssertionError):
        replay(bad, cfg)


def test_python_fri_prover_and_c_style_layers():
    """commit phase + grind + queries of the Python reading on a small random polynomial verify under the Python
    verifier: the two halves of the reading agree with each other (fold = interpolation at beta, coset shifts, X factor)"""
    rng = np.random.default_rng(11)
    log_n, rate_bits, cap_h = 6, 1, 1
    n = 1 << log_n
    cols = [[int(x) for x in _oracle.rand_field(rng, n)] for _ in range(3)]
    batch = g2.PolynomialBatch.from_values(cols, rate_bits, cap_h)
    ch = g2.Challenger()
    ch.observe_cap(batch.tree.cap)
    zeta = ch.get_ext()
    opened = [g2.eval_poly([g2.Ext(c) for c in p], zeta) for p in batch.polynomials]
    for o in opened:
        ch.observe_ext(o)
    alpha = ch.get_ext()
    comp = [g2.Ext(0)] * n
    for j, p in enumerate(batch.polynomials):
        aj = alpha ** j
        comp = [c + aj * x for c, x in zip(comp, p)]
    # (F(X) - F(zeta)) / (X - zeta), then times X
    quo, acc = [g2.Ext(0)] * n, g2.Ext(0)
    for k in range(n - 1, 0, -1):
        acc = comp[k] + acc * zeta
        quo[k - 1] = acc
    final = [g2.Ext(0)] + quo[:n - 1]
    arities = g2.reduction_arity_bits(2, 2, log_n, rate_bits, cap_h)
    assert arities == [2, 2]
    trees, final_poly = g2.fri_committed_trees(g2.lde(final, rate_bits), arities, rate_bits, cap_h, ch)
    w = g2.grind(ch, 0, 6)
    assert g2.pow_ok(g2.pow_response(ch, 0, w), 6)
    betas = None
    # the verifier derives the betas from the same transcript
    vch = g2.Challenger()
    vch.observe_cap(batch.tree.cap)
    assert vch.get_ext() == zeta
    for o in opened:
        vch.observe_ext(o)
    assert vch.get_ext() == alpha
    betas = []
    for t in trees:
        vch.observe_cap(t.cap)
        betas.append(vch.get_ext())
    for x in (0, 5, 77, 127):
        row = batch.tree.leaves[x]
        steps, xi = [], x
        for t, ab in zip(trees, arities):
            xi >>= ab
            ev = t.leaves[xi]
            steps.append([g2.Ext(ev[2 * k], ev[2 * k + 1]) for k in range(1 << ab)])
        err = g2.fri_verify_query(x, log_n, rate_bits, arities, alpha, [(zeta, row, opened)], [row], steps, betas, final_poly)
        assert err is None, (x, err)
