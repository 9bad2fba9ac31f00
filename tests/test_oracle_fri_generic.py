"""oracle/fri.c: PolynomialBatch::prove_openings / verify_fri_proof for ARBITRARY FriParams -- blowup 2 / 4 / 8, reduction
arities 2 .. 16 (constant or mixed), cap heights, both proof-of-work rules, salted (blinded) oracles -- on random polynomial
batches: the proofs verify, tampering is refused, and the independent Python reading (oracle/py/plonky2_generic.py)
verifies the C proofs query by query.  Also: the STARK oracle itself under non-default blowup / arity."""
import numpy as np
import pytest

from oracle.py import plonky2_generic as g2
from tests import _oracle

P = _oracle.P


def random_instance(seed, log_n, rate_bits, cap_height, ncols=(5, 3), salted=(False, True)):
    rng = np.random.default_rng(seed)
    n, m = 1 << log_n, 1 << (log_n + rate_bits)
    oracles = []
    for k, nc in enumerate(ncols):
        vals = _oracle.rand_field(rng, (nc, n))
        salt = _oracle.rand_field(rng, (4, m)) if salted[k] else None
        oracles.append(_oracle.SaltedBatch(vals, log_n, rate_bits, cap_height, from_values=(k == 0), salt=salt))
    zeta = tuple(int(x) for x in _oracle.rand_field(rng, 2))
    w = pow(1753635133440165772, 1 << (32 - log_n), P)
    gz = (zeta[0] * w % P, zeta[1] * w % P)
    # batch 0: everything at zeta; batch 1: a sub-range of oracle 0 and all of oracle 1 at g zeta (like zs / next-row openings)
    batches = [(zeta, [(0, 0, ncols[0]), (1, 0, ncols[1])]), (gz, [(0, 1, ncols[0] - 1), (1, 0, ncols[1])])]
    return oracles, batches


CONFIGS = [
    # log_n, rate_bits, cap_height, arity_bits, final_poly_bits, pow_rule, queries
    (8, 1, 4, 4, 3, 0, 7),
    (8, 2, 2, 3, 2, 0, 6),
    (9, 3, 4, 4, 5, 1, 5),      # the outer circuit's blowup 8
    (7, 1, 0, 1, 2, 0, 5),      # arity 2: 4-word leaves are not hashed (hash_or_noop)
    (8, 3, 3, 2, 4, 0, 5),
]


@pytest.mark.parametrize("log_n,rate_bits,cap_height,arity_bits,final_poly_bits,pow_rule,nq", CONFIGS)
def test_generic_opening_proofs_verify_and_reject_tampering(log_n, rate_bits, cap_height, arity_bits, final_poly_bits, pow_rule, nq):
    oracles, batches = random_instance(log_n * 10 + rate_bits, log_n, rate_bits, cap_height)
    fp = _oracle.fri_params(rate_bits=rate_bits, cap_height=cap_height, pow_bits=8, num_queries=nq, pow_rule=pow_rule, hiding=1,
                            arity_bits=arity_bits, final_poly_bits=final_poly_bits, degree_bits=log_n)
    assert fp.n_rounds >= 1
    seed = [3, 1, 4, 1, 5]
    pf = _oracle.fri_prove_openings(oracles, batches, log_n, fp, _oracle.challenger(seed))
    caps = [o.cap for o in oracles]
    args = ([o.ncols for o in oracles], [o.n_salt for o in oracles], batches, log_n, fp)
    assert _oracle.fri_verify_openings(pf, caps, *args, _oracle.challenger(seed)) == 0
    assert _oracle.fri_verify_openings(pf, caps, *args, _oracle.challenger(seed + [9])) != 0      # another transcript
    for pos in (8 + 3, len(pf) // 2, len(pf) - 3):                                              # an opened value, a query word, a path
        bad = pf.copy()
        bad[pos] ^= 1
        assert _oracle.fri_verify_openings(bad, caps, *args, _oracle.challenger(seed)) != 0, pos
    # mixed arities (FriReductionStrategy::Fixed)
    if log_n >= 8 and rate_bits <= 2:
        fp2 = _oracle.fri_params(rate_bits=rate_bits, cap_height=min(cap_height, 2), pow_bits=6, num_queries=4, hiding=1, arities=[3, 1, 2])
        o2, b2 = random_instance(77, log_n, rate_bits, min(cap_height, 2))
        pf2 = _oracle.fri_prove_openings(o2, b2, log_n, fp2, _oracle.challenger())
        assert _oracle.fri_verify_openings(pf2, [o.cap for o in o2], [o.ncols for o in o2], [o.n_salt for o in o2], b2, log_n, fp2,
                                           _oracle.challenger()) == 0


def test_python_reading_verifies_generic_c_proofs():
    log_n, rate_bits, cap_height, nq = 7, 2, 2, 5
    oracles, batches = random_instance(5, log_n, rate_bits, cap_height)
    fp = _oracle.fri_params(rate_bits=rate_bits, cap_height=cap_height, pow_bits=7, num_queries=nq, hiding=1, arity_bits=3,
                            final_poly_bits=1, degree_bits=log_n)
    arities = [fp.arity_bits[i] for i in range(fp.n_rounds)]
    assert arities == g2.reduction_arity_bits(3, 1, log_n, rate_bits, cap_height)
    pf = [int(x) for x in _oracle.fri_prove_openings(oracles, batches, log_n, fp, _oracle.challenger([42]))]
    # salted PolynomialBatch of the Python reading: same caps
    for o in oracles:
        coeffs = [[int(x) for x in row] for row in o.coeffs]
        pb = g2.PolynomialBatch(coeffs, rate_bits, cap_height)
        if o.n_salt:
            m = 1 << (log_n + rate_bits)
            rows = [pb.tree.leaves[j] + [int(o.salt[s][g2.reverse_bits(j, log_n + rate_bits)]) for s in range(4)] for j in range(m)]
            tree = g2.MerkleTree(rows, cap_height)
        else:
            tree = pb.tree
        assert tree.cap == [[int(x) for x in d] for d in o.cap]
    # replay
    pos = [8]

    def take(k):
        v = pf[pos[0]:pos[0] + k]
        pos[0] += k
        return v
    ch = g2.Challenger()
    ch.observe(42)
    opened = []
    for pt, ranges in batches:
        k = sum(e - b for _, b, e in ranges)
        vals = [g2.Ext(*take(2)) for _ in range(k)]
        for v in vals:
            ch.observe_ext(v)
        opened.append(vals)
    alpha = ch.get_ext()
    caps, betas = [], []
    for _ in arities:
        caps.append([take(4) for _ in range(1 << cap_height)])
        ch.observe_cap(caps[-1])
        betas.append(ch.get_ext())
    flen = (1 << log_n) >> sum(arities)
    final_poly = [g2.Ext(*take(2)) for _ in range(flen)]
    for c in final_poly:
        ch.observe_ext(c)
    w = take(1)[0]
    assert g2.pow_ok(g2.pow_response(ch, 0, w), 7)
    log_m = log_n + rate_bits
    for _ in range(nq):
        x = ch.get() % (1 << log_m)
        rows = []
        for o in oracles:
            row, sib = take(o.ncols + o.n_salt), [take(4) for _ in range(log_m - cap_height)]
            assert g2.verify_merkle_proof_to_cap(row, x, [[int(v) for v in d] for d in o.cap], sib)
            rows.append(row)
        steps, xi = [], x
        for r, ab in enumerate(arities):
            ev = take(2 << ab)
            xi >>= ab
            ns = max(0, log_m - sum(arities[:r + 1]) - cap_height)
            sib = [take(4) for _ in range(ns)]
            assert g2.verify_merkle_proof_to_cap(ev, xi, caps[r], sib)
            steps.append([g2.Ext(ev[2 * k], ev[2 * k + 1]) for k in range(1 << ab)])
        fb = []
        for (pt, ranges), vals in zip(batches, opened):
            at_x = [rows[o][c] for o, b, e in ranges for c in range(b, e)]          # salt words never enter (unsalted_eval)
            fb.append((g2.Ext(*pt), at_x, vals))
        assert g2.fri_verify_query(x, log_n, rate_bits, arities, alpha, fb, rows, steps, betas, final_poly) is None
    assert pos[0] == len(pf)


@pytest.mark.parametrize("rate_bits,arity_bits,final_poly_bits", [(2, 4, 5), (3, 3, 4), (1, 2, 3), (2, 1, 6)])
def test_stark_oracle_under_other_blowups_and_arities(rate_bits, arity_bits, final_poly_bits):
    """the three-oracle STARK prover / verifier with blowup 4 / 8 and arity 2 / 4 / 8 (quotient still on the 2N coset:
    the first 2N leaves of the larger LDE)"""
    ios = np.load("tests/golden/sipp_n4_ios.npz")["g1"]
    cfg = _oracle.default_config()
    cfg.rate_bits, cfg.arity_bits, cfg.final_poly_bits, cfg.num_queries, cfg.pow_bits = rate_bits, arity_bits, final_poly_bits, 10, 8
    pf = _oracle.stark_prove(0, ios, cfg)
    assert _oracle.stark_verify(pf, cfg) == 0
    bad = pf.copy()
    bad[len(pf) // 2] ^= 1
    assert _oracle.stark_verify(bad, cfg) != 0
    assert _oracle.stark_verify(pf, _oracle.default_config()) != 0
