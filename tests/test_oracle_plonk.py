"""plonky2's wire permutation argument (oracle/plonk.c: Z and partial products, the permutation terms of the vanishing polynomial,
quotient chunks, the verifier's check at zeta) on synthetic wires / sigmas -- SURVEY.md section 8f rank 2, the protocol-generic part of
`data.prove` (reference src/verifier_circuit.rs:253) that needs no circuit.  CPU only."""
import numpy as np
import pytest

from tests import _oracle

P = _oracle.P


def fri(log_n, rate_bits=3, cap_height=2, nq=6, arity=3, fpb=3):
    return _oracle.fri_params(rate_bits=rate_bits, cap_height=cap_height, pow_bits=6, num_queries=nq, pow_rule=0, hiding=0, arity_bits=arity,
                              final_poly_bits=fpb, degree_bits=log_n)


@pytest.mark.parametrize("log_n,R,D,C", [(6, 80, 8, 2), (5, 13, 4, 3), (7, 9, 2, 1), (6, 16, 8, 2)])
def test_partial_products_close_and_the_proof_verifies(log_n, R, D, C):
    """Z starts at 1 and every chunk relation holds on every row (the column order Z_0 .. Z_{C-1}, then the partial products per
    challenge); the whole flow proves and verifies for 80 routed wires in chunks of 8 (standard_recursion_config: 9 partial products
    per challenge, blowup 8), for a ragged last chunk (13 wires in chunks of 4), chunk size 2, and one or three challenges"""
    p = _oracle.plonk_params(R, D, C)
    wires, sig, perm = _oracle.plonk_random_instance(40 + log_n + R, log_n, R)
    n = 1 << log_n
    rng = np.random.default_rng(1)
    betas, gammas = _oracle.rand_field(rng, (C,)), _oracle.rand_field(rng, (C,))
    zs = _oracle.plonk_zs(wires, sig, log_n, p, betas, gammas)
    npd = _oracle.plonk_num_prods(p)
    assert zs.shape == (C * (1 + npd), n) and (zs[:C, 0] == 1).all()
    # python big-int check of row 3 of challenge C - 1: prev * prod(num) == next * prod(den) chunk by chunk, Z(g x) closes the row
    c, i = C - 1, 3
    w = pow(7, (P - 1) >> log_n, P)
    x = pow(w, i, P)
    accs = [int(zs[c, i])] + [int(zs[C + c * npd + q, i]) for q in range(npd)] + [int(zs[c, (i + 1) % n])]
    for q in range(npd + 1):
        num = den = 1
        for j in range(q * D, min((q + 1) * D, R)):
            num = num * ((int(wires[j, i]) + int(betas[c]) * pow(7, j, P) * x + int(gammas[c])) % P) % P
            den = den * ((int(wires[j, i]) + int(betas[c]) * int(sig[j, i]) + int(gammas[c])) % P) % P
        assert accs[q] * num % P == accs[q + 1] * den % P, q
    fp = fri(log_n, rate_bits=max(3, D.bit_length() - 1))
    pf = _oracle.plonk_perm_prove(wires, sig, log_n, p, fp)
    sig_cap = _oracle.Batch(sig, log_n, rate_bits=fp.rate_bits, cap_height=fp.cap_height).cap
    assert _oracle.plonk_perm_verify(pf, sig_cap, p, fp) == 0
    # tampering: an opened value, a cap word, another circuit digest
    bad = pf.copy()
    bad[8 + 3 * 4 * (1 << fp.cap_height) + 8 + 5] ^= 1
    assert _oracle.plonk_perm_verify(bad, sig_cap, p, fp) != 0
    bad = pf.copy()
    bad[9] ^= 1
    assert _oracle.plonk_perm_verify(bad, sig_cap, p, fp) != 0
    assert _oracle.plonk_perm_verify(pf, sig_cap, p, fp, digest=(1, 2, 3, 5)) != 0


def test_a_broken_copy_constraint_is_refused():
    """one wire changed so that two positions of a cycle disagree: Z does not return to 1 and the verifier's identity at zeta fails"""
    log_n, R, D, C = 6, 16, 8, 2
    p = _oracle.plonk_params(R, D, C)
    wires, sig, perm = _oracle.plonk_random_instance(7, log_n, R, n_cycles=40)
    fp = fri(log_n)
    sig_cap = _oracle.Batch(sig, log_n, rate_bits=fp.rate_bits, cap_height=fp.cap_height).cap
    assert _oracle.plonk_perm_verify(_oracle.plonk_perm_prove(wires, sig, log_n, p, fp), sig_cap, p, fp) == 0
    bad = wires.copy()
    pos = int(np.nonzero(perm != np.arange(perm.size))[0][0])        # a position inside a cycle of length > 1
    bad.reshape(-1)[pos] = (int(bad.reshape(-1)[pos]) + 1) % P
    assert _oracle.plonk_perm_verify(_oracle.plonk_perm_prove(bad, sig, log_n, p, fp), sig_cap, p, fp) == -210


def test_python_reading_replays_a_c_oracle_proof():
    """the second, independent reading (oracle/py/plonky2_generic.py, written in the shape of the Rust code): from the same wires and
    sigmas it derives the same challenges, the same Z / partial-product COLUMNS in the same order (its zs_partial_products cap equals
    the one in the C oracle's proof), and the verifier's identity at zeta holds on the proof's opened values with ITS evaluation of
    the vanishing terms"""
    from oracle.py import plonky2_generic as g2
    log_n, R, D, C = 4, 5, 2, 2
    p = _oracle.plonk_params(R, D, C)
    wires, sig, _ = _oracle.plonk_random_instance(11, log_n, R)
    fp = fri(log_n, rate_bits=2, cap_height=1, nq=3, arity=1, fpb=2)
    pf = [int(v) for v in _oracle.plonk_perm_prove(wires, sig, log_n, p, fp, digest=(9, 8, 7, 6))]
    cap_words = 4 << fp.cap_height
    wcap, zcap, qcap = pf[8:8 + cap_words], pf[8 + cap_words:8 + 2 * cap_words], pf[8 + 2 * cap_words:8 + 3 * cap_words]
    W = [[int(v) for v in row] for row in wires]
    S = [[int(v) for v in row] for row in sig]
    wb = g2.PolynomialBatch.from_values(W, fp.rate_bits, fp.cap_height)
    assert sum(wb.tree.cap, []) == wcap
    ch = g2.Challenger()
    ch.observe_many([9, 8, 7, 6])
    ch.observe_many([0, 0, 0, 0])
    ch.observe_cap(wb.tree.cap)
    betas, gammas = ch.get_n(C), ch.get_n(C)
    per_ch = [g2.wires_permutation_partial_products_and_zs(W, S, betas[i], gammas[i], D) for i in range(C)]
    zs_cols = [cols[-1] for cols in per_ch] + [c for cols in per_ch for c in cols[:-1]]       # Z first, then the partial products
    zb = g2.PolynomialBatch.from_values(zs_cols, fp.rate_bits, fp.cap_height)
    assert sum(zb.tree.cap, []) == zcap
    ch.observe_cap(zb.tree.cap)
    alphas = ch.get_n(C)
    ch.observe_cap([qcap[4 * k:4 * k + 4] for k in range(1 << fp.cap_height)])
    zeta = ch.get_ext()
    npd = _oracle.plonk_num_prods(p)
    op = pf[8 + 3 * cap_words + 8:]
    take = iter(range(0, 10 ** 9, 2))
    ext_at = lambda: (lambda k: g2.Ext(op[k], op[k + 1]))(next(take))
    sg_o = [ext_at() for _ in range(R)]
    w_o = [ext_at() for _ in range(R)]
    zs_o = [ext_at() for _ in range(C)]
    pp_o = [ext_at() for _ in range(C * npd)]
    q_o = [ext_at() for _ in range(C * D)]
    zn_o = [ext_at() for _ in range(C)]
    # the opened values ARE the polynomials at zeta / g zeta (python evaluation of its own coefficient columns)
    assert zs_o[1] == g2.eval_poly([g2.ext(c) for c in zb.polynomials[1]], zeta)
    assert zn_o[0] == g2.eval_poly([g2.ext(c) for c in zb.polynomials[0]], zeta * g2.ext(g2.primitive_root_of_unity(log_n)))
    van = g2.eval_vanishing_poly_permutation(log_n, zeta, w_o, sg_o, zs_o, zn_o, pp_o, betas, gammas, alphas, D)
    zeta_n = zeta ** (1 << log_n)
    for c in range(C):
        acc = g2.ext(0)
        for d in reversed(range(D)):
            acc = acc * zeta_n + q_o[c * D + d]
        assert van[c] == (zeta_n - g2.ext(1)) * acc, c


# ---- round 4: the circuit's gate-constraint terms in the quotient, the public inputs in the transcript (everything of prove() but the
# gates themselves and witness generation) ----
@pytest.mark.parametrize("log_n,R,D,C,K,n_pi", [(6, 16, 8, 2, 3, 5), (5, 13, 4, 3, 4, 0), (6, 9, 2, 1, 2, 9)])
def test_gate_terms_and_public_inputs_prove_and_verify(log_n, R, D, C, K, n_pi):
    """the synthetic circuit (K product gates on every row + a random wire permutation): the quotient with the gates' terms folded behind
    the permutation terms is a polynomial of the right degree (the verifier's identity at zeta holds with ITS evaluation of the gates on
    the opened wires); a wire that breaks ONE gate on ONE row is refused; so is a changed public input (it moves every challenge)"""
    p = _oracle.plonk_params(R, D, C)
    wires, sig, perm = _oracle.plonk_gate_instance(70 + log_n + R, log_n, R, K)
    fp = fri(log_n, rate_bits=max(3, D.bit_length() - 1))
    pis = [(7 * k + 1) % P for k in range(n_pi)]
    digest = (5, 6, 7, 8)
    pf = _oracle.plonk_prove_ex(wires, sig, log_n, p, fp, digest, pis, K)
    assert int(pf[0]) == 0x324b4c5050504953 and int(pf[6]) == K and int(pf[7]) == n_pi and int(pf[5]) == len(pf)
    assert [int(v) for v in pf[len(pf) - n_pi:]] == pis
    sig_cap = _oracle.Batch(sig, log_n, rate_bits=fp.rate_bits, cap_height=fp.cap_height).cap
    assert _oracle.plonk_verify_ex(pf, sig_cap, p, fp, digest, K) == 0
    # the permutation-only verifier / another gate count do not take this proof
    assert _oracle.plonk_perm_verify(pf, sig_cap, p, fp, digest=digest) != 0
    assert _oracle.plonk_verify_ex(pf, sig_cap, p, fp, digest, K - 1) != 0
    if n_pi:
        bad = pf.copy()
        bad[-1] = (int(bad[-1]) + 1) % P
        assert _oracle.plonk_verify_ex(bad, sig_cap, p, fp, digest, K) != 0
    # one product gate broken on one row (the output column is unrouted: the permutation argument alone would not notice)
    broken = wires.copy()
    broken[2, 3] = (int(broken[2, 3]) + 1) % P
    assert _oracle.plonk_perm_verify(_oracle.plonk_perm_prove(broken, sig, log_n, p, fp), sig_cap, p, fp) == 0
    assert _oracle.plonk_verify_ex(_oracle.plonk_prove_ex(broken, sig, log_n, p, fp, digest, pis, K), sig_cap, p, fp, digest, K) == -210


def test_gate_terms_vanish_on_the_subgroup_and_enter_behind_the_permutation_terms():
    """the terms the oracle folds: zero on the trace domain (every D-th point of the coset after un-shifting is not on it, so check by
    interpolation: term polynomial = product of wire polynomials mod nothing, degree < 2N), and their position in reduce_with_powers --
    with alpha = 0 only the FIRST term (Z(1) = 1) survives, with the gate terms scaled by alpha^(C + C m + k)"""
    log_n, R, D, C, K = 5, 7, 2, 1, 2
    p = _oracle.plonk_params(R, D, C)
    wires, sig, _ = _oracle.plonk_gate_instance(3, log_n, R, K)
    n = 1 << log_n
    rng = np.random.default_rng(9)
    betas, gammas, alphas = (_oracle.rand_field(rng, (C,)) for _ in range(3))
    wb = _oracle.Batch(wires, log_n, rate_bits=1, cap_height=0)
    sb = _oracle.Batch(sig, log_n, rate_bits=1, cap_height=0)
    zs = _oracle.plonk_zs(wires, sig, log_n, p, betas, gammas)
    zb = _oracle.Batch(zs, log_n, rate_bits=1, cap_height=0)
    gt = _oracle.plonk_gate_terms_coset(wb.coeffs, log_n, 1, K)
    q0 = _oracle.plonk_quotient_chunks(wb.coeffs, sb.coeffs, zb.coeffs, log_n, p, betas, gammas, alphas)
    q1 = _oracle.plonk_quotient_chunks_ex(wb.coeffs, sb.coeffs, zb.coeffs, log_n, p, betas, gammas, alphas, gt)
    # honest wires: both are polynomial quotients, and they differ by alpha^(n_perm + k) * (gate term k / Z_H), itself a polynomial
    npd = _oracle.plonk_num_prods(p)
    n_perm = C + C * (npd + 1)
    a = int(alphas[0])
    w2 = pow(7, (P - 1) >> (log_n + 1), P)
    x = 7 * pow(w2, 5, P) % P                                   # a point of the coset: evaluate both sides there
    ev = lambda chunks: sum(int(c) * pow(x, j, P) for j, c in enumerate(np.concatenate(list(chunks)))) % P
    zh = (pow(x, n, P) - 1) % P
    want = (ev(q0[:D]) + sum(pow(a, n_perm + k, P) * int(gt[k, 5]) for k in range(K)) * pow(zh, P - 2, P)) % P
    assert ev(q1[:D]) == want


def test_python_reading_replays_a_proof_with_gates_and_public_inputs():
    """the second reading (oracle/py/plonky2_generic.py) on a "SIPPPLK2" proof: hash_n_to_hash_no_pad of the public inputs, the transcript,
    and the verifier's identity at zeta with the gate terms it computes itself from the opened wires"""
    from oracle.py import plonky2_generic as g2
    log_n, R, D, C, K = 4, 7, 2, 2, 2
    p = _oracle.plonk_params(R, D, C)
    wires, sig, _ = _oracle.plonk_gate_instance(21, log_n, R, K)
    fp = fri(log_n, rate_bits=2, cap_height=1, nq=3, arity=1, fpb=2)
    pis = [3, 1, 4, 1, 5, 9, 2, 6, 5, 3]
    pf = [int(v) for v in _oracle.plonk_prove_ex(wires, sig, log_n, p, fp, (9, 8, 7, 6), pis, K)]
    cap_words = 4 << fp.cap_height
    wcap, zcap, qcap = pf[8:8 + cap_words], pf[8 + cap_words:8 + 2 * cap_words], pf[8 + 2 * cap_words:8 + 3 * cap_words]
    assert pf[len(pf) - len(pis):] == pis
    ch = g2.Challenger()
    ch.observe_many([9, 8, 7, 6])
    ch.observe_many(g2.hash_no_pad(pis))
    ch.observe_cap([wcap[4 * k:4 * k + 4] for k in range(1 << fp.cap_height)])
    betas, gammas = ch.get_n(C), ch.get_n(C)
    ch.observe_cap([zcap[4 * k:4 * k + 4] for k in range(1 << fp.cap_height)])
    alphas = ch.get_n(C)
    ch.observe_cap([qcap[4 * k:4 * k + 4] for k in range(1 << fp.cap_height)])
    zeta = ch.get_ext()
    npd = _oracle.plonk_num_prods(p)
    op = pf[8 + 3 * cap_words + 8:]
    take = iter(range(0, 10 ** 9, 2))
    ext_at = lambda: (lambda k: g2.Ext(op[k], op[k + 1]))(next(take))
    sg_o = [ext_at() for _ in range(R)]
    w_o = [ext_at() for _ in range(R)]
    zs_o = [ext_at() for _ in range(C)]
    pp_o = [ext_at() for _ in range(C * npd)]
    q_o = [ext_at() for _ in range(C * D)]
    zn_o = [ext_at() for _ in range(C)]
    gates = [w_o[3 * k] * w_o[3 * k + 1] - w_o[3 * k + 2] for k in range(K)]
    van = g2.eval_vanishing_poly_permutation(log_n, zeta, w_o, sg_o, zs_o, zn_o, pp_o, betas, gammas, alphas, D, gates)
    zeta_n = zeta ** (1 << log_n)
    for c in range(C):
        acc = g2.ext(0)
        for d in reversed(range(D)):
            acc = acc * zeta_n + q_o[c * D + d]
        assert van[c] == (zeta_n - g2.ext(1)) * acc, c
    # without the gate terms the identity must NOT hold (they really are in the quotient)
    van0 = g2.eval_vanishing_poly_permutation(log_n, zeta, w_o, sg_o, zs_o, zn_o, pp_o, betas, gammas, alphas, D)
    assert van0[0] != van[0]


# ---------------------------------------------------------------------------------------------------- gates as data (round 5)
def _synth(log_n, num_wires=40, num_routed=24, seed=5, pis=(3, 1, 4, 1, 5)):
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import plonk_synth as ps
    circ = ps.circuit(num_wires, num_routed)
    pih = [int(x) for x in _oracle.hash_no_pad(np.array(pis, dtype=np.uint64))]
    wires, cs, gate = ps.witness(circ, log_n, seed, pih)
    return ps, circ, wires, cs, gate, list(pis), pih


def test_synthetic_gate_circuit_witness_satisfies_its_programs():
    """tools/plonk_synth.py: every row satisfies the gate its selector names, under three readings of the programs (plain Python over the
    integers, oracle/plonk_gates.c over the base field, oracle/py/plonky2_generic.py over the extension); on a row of ANOTHER gate the
    filter is zero there and nowhere else; a tampered cell shows in exactly the constraints that read it"""
    from oracle.py import plonky2_generic as g2
    ps, circ, wires, cs, gate, pis, pih = _synth(7)
    n = 1 << 7
    assert ps.check_rows(circ, wires, cs, pih, range(n))
    assert set(int(g) for g in gate) == {0, 1, 2, 3, 4}
    for r in (0, 1, 2, 3, 4, 6, 77):
        out = _oracle.plonk_gate_constraints_base(circ, wires[:, r], cs[:circ["num_constants"], r], pih)
        assert not out.any(), (r, int(gate[r]))
        e = g2.evaluate_gate_constraints(circ["gates"], circ["programs"], circ["num_selectors"], [g2.ext(int(v)) for v in wires[:, r]],
                                         [g2.ext(int(v)) for v in cs[:circ["num_constants"], r]], pih)
        assert all(v == g2.ext(0) for v in e)
    r = int(np.flatnonzero(gate == 1)[0])
    bad = wires[:, r].copy()
    bad[3] ^= 1                                           # the output of arithmetic op 0
    out = _oracle.plonk_gate_constraints_base(circ, bad, cs[:circ["num_constants"], r], pih)
    assert out[0] != 0 and not out[1:].any()
    e = g2.evaluate_gate_constraints(circ["gates"], circ["programs"], circ["num_selectors"], [g2.ext(int(v)) for v in bad],
                                     [g2.ext(int(v)) for v in cs[:circ["num_constants"], r]], pih)
    assert int(e[0][0]) == int(out[0]) and int(e[0][1]) == 0


def test_recursion_shaped_circuit_witness_proof_and_rejection():
    """round 6: at the column counts of CircuitConfig::standard_ecc_config the synthetic circuit is recursion-shaped:
    Poseidon (the whole permutation: 118 constraints of degree 7), U32 multiply-add with 2-bit range checks, random access, reducing,
    arithmetic, base-sum, constant, public-input gates in three selector groups.  Every row satisfies its gate under the three readings of
    the programs; a Poseidon row's out wires ARE the permutation of its in wires; the oracle proves and verifies; a tampered S-box wire
    of a partial round is refused"""
    from oracle.py import plonky2_generic as g2
    ps, circ, wires, cs, gate, pis, pih = _synth(7, 136, 80, seed=9)
    assert circ["rich"] and circ["num_gate_constraints"] == 118 and circ["num_selectors"] == 3 and circ["num_constants"] == 5
    assert [g[5] for g in circ["gates"]] == [0, 20, 33, 4, 2, 105, 40, 80, 118]
    n = 1 << 7
    assert set(int(g) for g in gate) == set(range(9))
    assert ps.check_rows(circ, wires, cs, pih, range(n))
    for kind in range(9):
        r = int(np.flatnonzero(gate == kind)[0])
        out = _oracle.plonk_gate_constraints_base(circ, wires[:, r], cs[:5, r], pih)
        assert not out.any(), kind
        e = g2.evaluate_gate_constraints(circ["gates"], circ["programs"], 3, [g2.ext(int(v)) for v in wires[:, r]], [g2.ext(int(v)) for v in cs[:5, r]], pih)
        assert all(v == g2.ext(0) for v in e), kind
    r = int(np.flatnonzero(gate == 8)[1])
    assert [int(wires[12 + i, r]) for i in range(12)] == g2.poseidon([int(wires[i, r]) for i in range(12)])
    # U32 rows: a b + c = lo + 2^32 hi over the integers, limbs are the 2-bit digits
    r = int(np.flatnonzero(gate == 5)[0])
    a, b, c, lo, hi = (int(wires[k, r]) for k in range(5))
    assert a * b + c == lo + (hi << 32) and sum(int(wires[5 + i, r]) << (2 * i) for i in range(16)) == lo
    p = _oracle.plonk_params(80, 8, 2)
    fp = fri(7, rate_bits=3, cap_height=2, nq=4, arity=2, fpb=3)
    digest = (9, 8, 7, 6)
    pf = _oracle.plonk_prove_gates(wires, cs, 7, p, fp, circ, digest, pis)
    cs_cap = _oracle.Batch(cs, 7, rate_bits=3, cap_height=2).cap
    assert int(pf[10]) == 118 and _oracle.plonk_verify_gates(pf, cs_cap, p, fp, circ, digest) == 0
    w2 = wires.copy()
    w2[ps.poseidon_sbox_wire(10, 0), int(np.flatnonzero(gate == 8)[3])] ^= 1      # the S-box input of a partial round
    pf2 = _oracle.plonk_prove_gates(w2, cs, 7, p, fp, circ, digest, pis)
    assert _oracle.plonk_verify_gates(pf2, cs_cap, p, fp, circ, digest) == -210
    w3 = wires.copy()
    w3[5 + 3, int(np.flatnonzero(gate == 5)[1])] = 4                                # a "2-bit" limb holding 4: the range product is not zero
    pf3 = _oracle.plonk_prove_gates(w3, cs, 7, p, fp, circ, digest, pis)
    assert _oracle.plonk_verify_gates(pf3, cs_cap, p, fp, circ, digest) == -210


def test_witness_generators_fill_what_the_numpy_generator_fills():
    """round 6: generate_partial_witness for the recursion-shaped gate set as row-local generators given as data (oracle/plonk_witness.c,
    the checker of sipp_plonk_generate_witness).  From the INPUT cells alone (every generated cell blanked; tools/plonk_synth.py names
    them) the C generators rebuild the numpy generator's wire table bit for bit; the inputs-only form of the numpy generator holds the
    same inputs; the result satisfies every gate program; a layout that leaves the table is refused"""
    ps, circ, wires, cs, gate, pis, pih = _synth(8, 136, 80, seed=21)
    K = circ["num_constants"]
    gens = ps.generators(circ)
    assert sorted(g[2] for g in gens) == list(range(1, 9))                  # every gate but the no-op has a generator
    blank = ps.blank_generated(circ, wires, gate, value=0x1234)
    assert int((blank != wires).sum()) > 15000
    got = _oracle.plonk_generate_witness(blank, cs[:K], 8, gens, pih)
    assert (got == wires).all()
    assert ps.check_rows(circ, got, cs, pih, range(1 << 8))
    w_in, cs_in, gate_in = ps.witness(circ, 8, 21, pih, inputs_only=True)
    assert (cs_in == cs).all() and (gate_in == gate).all()
    assert (ps.blank_generated(circ, w_in, gate, value=0x1234) == blank).all()
    assert (_oracle.plonk_generate_witness(w_in, cs[:K], 8, gens, pih) == wires).all()
    # one family at a time touches only its rows' generated cells
    only = _oracle.plonk_generate_witness(blank, cs[:K], 8, [g for g in gens if g[0] == ps.GEN_POSEIDON], pih)
    rows8 = np.flatnonzero(gate == 8)
    assert (only[:, rows8] == wires[:, rows8]).all() and (np.delete(only, rows8, axis=1) == np.delete(blank, rows8, axis=1)).all()
    for bad in ((ps.GEN_POSEIDON, 2, 8, 0, 12, 40, 0, 0), (ps.GEN_ARITHMETIC, 0, 1, 35, 3, 4, 0, 0), (ps.GEN_CONSTANT, 0, 4, 2, 4, 0, 0, 0),
                (ps.GEN_REDUCING, 5, 7, 40, 7, 0, 0, 0), (9, 0, 1, 0, 0, 0, 0, 0)):
        with pytest.raises(RuntimeError):
            _oracle.plonk_generate_witness(blank, cs[:K], 8, [bad], pih)


def test_levelled_witness_generation_follows_copy_constraints_from_outputs_to_inputs():
    """the chained circuit (tools/plonk_synth.chain_schedule): Poseidon hash chains, arithmetic rows reading Poseidon outputs, reducing rows
    reading arithmetic outputs -- copy constraints (2-cycles of sigma) from outputs to inputs of OTHER rows, so generators have an order.
    oracle/plonk_witness.c runs the build-time schedule level by level and rebuilds the numpy generator's table bit for bit from the
    inputs alone; the row-local pass cannot; every gate program and every copy constraint holds; the oracle proves and verifies the
    circuit (sigma carries the new cycles: a proof exists only if the copies hold); a schedule that leaves the table is refused"""
    log_n, cl = 8, 6
    ps, circ, _w, _cs, _gate, pis, pih = _synth(log_n, 136, 80, seed=33)
    wires, cs, gate = ps.witness(circ, log_n, 33, pih, chain_len=cl)
    sc = ps.chain_schedule(log_n, cl)
    K, n = circ["num_constants"], 1 << log_n
    assert sc["n_levels"] >= cl and len(sc["copy_src"]) > 100 and sorted(sc["rows"].tolist()) == list(range(n))
    flat = wires.reshape(-1)
    assert (flat[sc["copy_src"].astype(np.int64)] == flat[sc["copy_dst"].astype(np.int64)]).all()
    assert ps.check_rows(circ, wires, cs, pih, range(n))
    blank = ps.blank_generated(circ, wires, gate, value=77, sched=sc)
    gens = ps.generators(circ)
    assert (_oracle.plonk_generate_witness_levels(blank, cs[:K], log_n, gens, pih, sc) == wires).all()
    assert not (_oracle.plonk_generate_witness(blank, cs[:K], log_n, gens, pih) == wires).all()
    w_in = ps.witness(circ, log_n, 33, pih, inputs_only=True, chain_len=cl)[0]
    assert (_oracle.plonk_generate_witness_levels(w_in, cs[:K], log_n, gens, pih, sc) == wires).all()
    p = _oracle.plonk_params(80, 8, 2)
    fp = fri(log_n, rate_bits=3, cap_height=2, nq=4, arity=2, fpb=3)
    digest = (5, 6, 7, 8)
    pf = _oracle.plonk_prove_gates(wires, cs, log_n, p, fp, circ, digest, pis)
    cs_cap = _oracle.Batch(cs, log_n, rate_bits=3, cap_height=2).cap
    assert _oracle.plonk_verify_gates(pf, cs_cap, p, fp, circ, digest) == 0
    # the witness of the row-local pass breaks copy constraints: its proof is refused
    pf2 = _oracle.plonk_prove_gates(_oracle.plonk_generate_witness(blank, cs[:K], log_n, gens, pih), cs, log_n, p, fp, circ, digest, pis)
    assert _oracle.plonk_verify_gates(pf2, cs_cap, p, fp, circ, digest) != 0
    for key, val in (("rows", n), ("copy_dst", 136 * n)):
        bad = dict(sc, **{key: sc[key].copy()})
        bad[key][3] = val
        with pytest.raises(RuntimeError):
            _oracle.plonk_generate_witness_levels(blank, cs[:K], log_n, gens, pih, bad)


def test_gates_as_data_prove_verify_and_reject():
    """orc_plonk_prove_gates / orc_plonk_verify_gates ("SIPPPLK3"): the whole outer flow with the gate set as data -- constants_sigmas
    (selectors, gate constants, sigmas), all wires, Z / partial products, quotient chunks with the gate terms, openings, FRI.  Accepted;
    refused after tampering with a wire cell of the witness (the quotient no longer fits), with an opened value, with a public input,
    and for another circuit's programs."""
    ps, circ, wires, cs, gate, pis, pih = _synth(7)
    log_n = 7
    p = _oracle.plonk_params(circ["num_routed"], 8, 2)
    fp = fri(log_n, rate_bits=3, cap_height=2, nq=4, arity=2, fpb=3)
    digest = (9, 8, 7, 6)
    pf = _oracle.plonk_prove_gates(wires, cs, log_n, p, fp, circ, digest, pis)
    assert int(pf[6]) == circ["num_wires"] and int(pf[7]) == 4 and int(pf[10]) == circ["num_gate_constraints"] and int(pf[11]) == len(pis)
    cs_cap = _oracle.Batch(cs, log_n, rate_bits=3, cap_height=2).cap
    assert _oracle.plonk_verify_gates(pf, cs_cap, p, fp, circ, digest) == 0
    bad = pf.copy()
    bad[16 + 3 * 16 + 8 + 2 * (4 + circ["num_routed"]) + 6] ^= 1      # an opened wire value
    assert _oracle.plonk_verify_gates(bad, cs_cap, p, fp, circ, digest) != 0
    bad = pf.copy()
    bad[-1] ^= 1                                                       # a public input: another hash, another transcript
    assert _oracle.plonk_verify_gates(bad, cs_cap, p, fp, circ, digest) != 0
    other = dict(circ, programs=circ["programs"].copy())
    other["programs"][1] = 2                                           # arithmetic op 0 with coefficient 2
    assert _oracle.plonk_verify_gates(pf, cs_cap, p, fp, other, digest) == -210
    w2 = wires.copy()
    r = int(np.flatnonzero(gate == 4)[3])
    w2[12 + 5, r] ^= 1                                                 # an S-box output: the constraint polynomial is no multiple of Z_H,
    pf2 = _oracle.plonk_prove_gates(w2, cs, log_n, p, fp, circ, digest, pis)     # the interpolated "quotient" does not reproduce it at zeta
    assert _oracle.plonk_verify_gates(pf2, cs_cap, p, fp, circ, digest) == -210


def test_python_reading_replays_a_proof_with_gates_as_data():
    """the second reading on a "SIPPPLK3" proof: transcript, the gate constraints at zeta from the OPENED constants and wires
    (oracle/py/plonky2_generic.py evaluate_gate_constraints), the verifier's identity"""
    from oracle.py import plonky2_generic as g2
    ps, circ, wires, cs, gate, pis, pih = _synth(6, num_wires=36, num_routed=16)
    log_n, R, D, C, K, W = 6, 16, 8, 2, 4, 36
    p = _oracle.plonk_params(R, D, C)
    fp = fri(log_n, rate_bits=3, cap_height=1, nq=3, arity=1, fpb=2)
    pf = [int(v) for v in _oracle.plonk_prove_gates(wires, cs, log_n, p, fp, circ, (1, 2, 3, 4), pis)]
    cap_words = 4 << fp.cap_height
    wcap, zcap, qcap = (pf[16 + k * cap_words:16 + (k + 1) * cap_words] for k in range(3))
    assert pf[len(pf) - len(pis):] == pis and g2.hash_no_pad(pis) == pih
    ch = g2.Challenger()
    ch.observe_many([1, 2, 3, 4])
    ch.observe_many(pih)
    ch.observe_cap([wcap[4 * k:4 * k + 4] for k in range(1 << fp.cap_height)])
    betas, gammas = ch.get_n(C), ch.get_n(C)
    ch.observe_cap([zcap[4 * k:4 * k + 4] for k in range(1 << fp.cap_height)])
    alphas = ch.get_n(C)
    ch.observe_cap([qcap[4 * k:4 * k + 4] for k in range(1 << fp.cap_height)])
    zeta = ch.get_ext()
    npd = _oracle.plonk_num_prods(p)
    op = pf[16 + 3 * cap_words + 8:]
    take = iter(range(0, 10 ** 9, 2))
    ext_at = lambda: (lambda k: g2.Ext(op[k], op[k + 1]))(next(take))
    c_o = [ext_at() for _ in range(K)]
    sg_o = [ext_at() for _ in range(R)]
    w_o = [ext_at() for _ in range(W)]
    zs_o = [ext_at() for _ in range(C)]
    pp_o = [ext_at() for _ in range(C * npd)]
    q_o = [ext_at() for _ in range(C * D)]
    zn_o = [ext_at() for _ in range(C)]
    terms = g2.evaluate_gate_constraints(circ["gates"], circ["programs"], circ["num_selectors"], w_o, c_o, pih)
    van = g2.eval_vanishing_poly_permutation(log_n, zeta, w_o[:R], sg_o, zs_o, zn_o, pp_o, betas, gammas, alphas, D, terms)
    zeta_n = zeta ** (1 << log_n)
    for c in range(C):
        acc = g2.ext(0)
        for d in reversed(range(D)):
            acc = acc * zeta_n + q_o[c * D + d]
        assert van[c] == (zeta_n - g2.ext(1)) * acc, c
