"""plonky2's wire permutation argument through the C ABI (sipp_plonk_zs_partial_products, sipp_plonk_quotient_chunks,
sipp_plonk_perm_prove) against oracle/plonk.c: Z and partial-product columns, quotient coefficient chunks and the complete flat
proof identical word for word; the oracle's verifier accepts the device's proof.  SURVEY.md section 8f rank 2, the protocol-generic part
of `data.prove` (reference src/verifier_circuit.rs:253) that needs no circuit."""
import ctypes as C

import numpy as np
import pytest

from tests import _oracle, _verify
from tests.test_gpu_fri_generic import to_params
from tests.test_oracle_plonk import fri

P = _oracle.P

pytestmark = pytest.mark.gpu

# log_n, routed wires, chunk size (quotient degree factor), challenges, rate_bits
CONFIGS = [
    (10, 80, 8, 2, 3),      # standard_recursion_config: 80 routed wires, 9 partial products per challenge, blowup 8
    (12, 80, 8, 2, 3),
    (10, 13, 4, 3, 2),      # ragged last chunk, three challenges, blowup 4
    (11, 9, 2, 1, 1),       # chunk size 2, one challenge, blowup 2
    (10, 135, 8, 2, 3),     # all 135 wires of the standard configuration routed: 17 chunks
]


@pytest.fixture(scope="module")
def ctx():
    import sipp_amd
    c = sipp_amd.Ctx(workspace_bytes=3 << 30)
    yield c
    c.close()


def dev(a):
    from sipp_amd._lib import to_device
    return to_device(a)


def host(t):
    from sipp_amd._lib import to_host
    return to_host(t)


@pytest.mark.parametrize("log_n,R,D,C,rate_bits", CONFIGS)
def test_zs_quotient_and_whole_proof_match_the_oracle(ctx, log_n, R, D, C, rate_bits):
    import sipp_amd
    op = _oracle.plonk_params(R, D, C)
    gp = sipp_amd.PlonkParams(R, D, C)
    wires, sig, _ = _oracle.plonk_random_instance(900 + log_n + R, log_n, R)
    rng = np.random.default_rng(3)
    betas, gammas, alphas = (_oracle.rand_field(rng, (C,)) for _ in range(3))
    d_w, d_s = dev(wires), dev(sig)
    # Z and the partial products, every cell
    ref_zs = _oracle.plonk_zs(wires, sig, log_n, op, betas, gammas)
    got_zs = ctx.plonk_zs(d_w, d_s, log_n, gp, betas, gammas)
    assert (host(got_zs) == ref_zs).all()
    # quotient chunks from the committed LDEs against the oracle's from coefficients
    cap_h = 2
    _, _, (wc, wl, _) = ctx.commit_ex(d_w, log_n, rate_bits, cap_h)
    _, _, (sc, sl, _) = ctx.commit_ex(d_s, log_n, rate_bits, cap_h)
    _, _, (zc, zl, _) = ctx.commit_ex(got_zs, log_n, rate_bits, cap_h)
    ref_q = _oracle.plonk_quotient_chunks(host(wc), host(sc), host(zc), log_n, op, betas, gammas, alphas)
    got_q = host(ctx.plonk_quotient_chunks(wl, sl, zl, log_n, rate_bits, gp, betas, gammas, alphas))
    bad = np.argwhere(got_q != ref_q)
    assert bad.size == 0, "first mismatching (chunk, coefficient): %s" % bad[:4].tolist()
    # the whole argument: flat proof word for word, and the oracle's verifier on the device's proof
    fp = fri(log_n, rate_bits=rate_bits, cap_height=cap_h, nq=5, arity=3, fpb=3)
    ref = _oracle.plonk_perm_prove(wires, sig, log_n, op, fp, digest=(11, 12, 13, 14))
    got = ctx.plonk_perm_prove(d_w, d_s, log_n, gp, to_params(fp), digest=(11, 12, 13, 14))
    assert len(got) == len(ref)
    diff = np.nonzero(got != ref)[0]
    assert diff.size == 0, "first mismatch at word %d of %d" % (diff[0], len(ref))
    sig_cap = _oracle.Batch(sig, log_n, rate_bits=rate_bits, cap_height=cap_h).cap
    assert _oracle.plonk_perm_verify(got, sig_cap, op, fp, digest=(11, 12, 13, 14)) == 0


def test_broken_copy_constraint_gives_a_proof_the_verifier_refuses(ctx):
    """the device proves whatever it is given (as plonky2's prover does); a wire that breaks a copy constraint yields a proof whose
    identity at zeta fails in the verifier (-210), word for word the oracle's proof of the same broken witness"""
    import sipp_amd
    log_n, R, D, C, rate_bits = 10, 16, 8, 2, 3
    op, gp = _oracle.plonk_params(R, D, C), sipp_amd.PlonkParams(R, D, C)
    wires, sig, perm = _oracle.plonk_random_instance(5, log_n, R, n_cycles=300)
    pos = int(np.nonzero(perm != np.arange(perm.size))[0][0])
    wires.reshape(-1)[pos] = (int(wires.reshape(-1)[pos]) + 1) % _oracle.P
    fp = fri(log_n, rate_bits=rate_bits, cap_height=1, nq=4, arity=4, fpb=2)
    got = ctx.plonk_perm_prove(dev(wires), dev(sig), log_n, gp, to_params(fp))
    assert (got == _oracle.plonk_perm_prove(wires, sig, log_n, op, fp)).all()
    sig_cap = _oracle.Batch(sig, log_n, rate_bits=rate_bits, cap_height=1).cap
    assert _oracle.plonk_perm_verify(got, sig_cap, op, fp) == -210


def test_argument_errors(ctx):
    import sipp_amd
    w = dev(np.zeros((4, 1 << 10), dtype=np.uint64))
    for bad in (sipp_amd.PlonkParams(4, 3, 2), sipp_amd.PlonkParams(4, 1, 2), sipp_amd.PlonkParams(4, 2, 9), sipp_amd.PlonkParams(0, 2, 1),
                sipp_amd.PlonkParams(2 * 33, 2, 1)):
        with pytest.raises(sipp_amd.SippError):
            ctx.plonk_zs(w, w, 10, bad, [1] * 9, [2] * 9)
    with pytest.raises(sipp_amd.SippError):                      # blowup 2 cannot carry a quotient degree factor of 4
        ctx.plonk_quotient_chunks(w, w, w, 10, 1, sipp_amd.PlonkParams(4, 4, 1), [1], [2], [3])


def test_workspace_too_small_fails_cleanly_and_the_ctx_survives():
    """the whole-argument call allocates its four oracles from the ctx's arena: an arena that cannot hold them gives SIPP_E_NOMEM
    (no fault, no partial proof), and the same ctx proves a smaller instance afterwards"""
    import sipp_amd
    c = sipp_amd.Ctx(workspace_bytes=96 << 20)
    try:
        log_n, R, D, C = 13, 80, 8, 2
        op, gp = _oracle.plonk_params(R, D, C), sipp_amd.PlonkParams(R, D, C)
        wires, sig, _ = _oracle.plonk_random_instance(77, log_n, R)
        fp = fri(log_n, rate_bits=3, cap_height=2, nq=4, arity=4, fpb=3)
        with pytest.raises(sipp_amd.SippError) as e:
            c.plonk_perm_prove(dev(wires), dev(sig), log_n, gp, to_params(fp))
        assert e.value.code == -3
        log_n = 10
        wires, sig, _ = _oracle.plonk_random_instance(78, log_n, R)
        fp = fri(log_n, rate_bits=3, cap_height=2, nq=4, arity=4, fpb=3)
        got = c.plonk_perm_prove(dev(wires), dev(sig), log_n, gp, to_params(fp))
        assert (got == _oracle.plonk_perm_prove(wires, sig, log_n, op, fp)).all()
    finally:
        c.close()


@pytest.mark.parametrize("log_n,R,D,C,rate_bits,K,n_pi,precommitted", [(10, 80, 8, 2, 3, 20, 6, True), (11, 13, 4, 3, 2, 4, 0, False),
                                                                        (10, 9, 2, 1, 1, 3, 17, True)])
def test_gate_terms_and_public_inputs_match_the_oracle(ctx, log_n, R, D, C, rate_bits, K, n_pi, precommitted):
    """round 4 (SURVEY 8f rank 2 continued): everything of prove() except gate evaluation.  The caller's gate-constraint terms (here the
    synthetic circuit of oracle/plonk.h: K product gates, evaluated on the coset by the oracle's helper and handed to the device in LEAF
    order) are folded behind the permutation terms with the same powers of alpha; the public inputs are hashed into the transcript and
    travel with the proof; the wires / sigmas oracles may be committed by the caller beforehand.  Quotient chunks and the whole "SIPPPLK2"
    proof word for word, the oracle's verifier (which evaluates the gates at zeta itself) accepts the device's proof."""
    import sipp_amd
    op = _oracle.plonk_params(R, D, C)
    gp = sipp_amd.PlonkParams(R, D, C)
    wires, sig, _ = _oracle.plonk_gate_instance(1300 + log_n + R, log_n, R, K)
    rng = np.random.default_rng(5)
    betas, gammas, alphas = (_oracle.rand_field(rng, (C,)) for _ in range(3))
    d_w, d_s = dev(wires), dev(sig)
    cap_h = 2
    log_d = D.bit_length() - 1
    n, m = 1 << log_n, (1 << log_n) << rate_bits
    w_or, w_cap, (wc, wl, wt) = ctx.commit_ex(d_w, log_n, rate_bits, cap_h)
    s_or, _, (sc, sl, st) = ctx.commit_ex(d_s, log_n, rate_bits, cap_h)
    got_zs = ctx.plonk_zs(d_w, d_s, log_n, gp, betas, gammas)
    _, _, (zc, zl, _) = ctx.commit_ex(got_zs, log_n, rate_bits, cap_h)
    # the gates' terms: natural order on the coset 7 <w_{N D}> from the oracle -> leaf order of the blowup-2^rate_bits LDE (the quotient
    # coset is its first N D leaves: leaf j = natural row bitrev(j) of the coset)
    gt_nat = _oracle.plonk_gate_terms_coset(host(wc), log_n, log_d, K)
    nd, lq = n << log_d, log_n + log_d
    rev = np.array([int(format(j, "0%db" % lq)[::-1], 2) for j in range(nd)])
    gt_leaf = np.zeros((K, m), dtype=np.uint64)
    gt_leaf[:, :nd] = gt_nat[:, rev]
    if rate_bits > log_d:
        gt_leaf[:, nd:] = 0xDEADBEEF                    # rows outside the quotient coset are never read
    d_gt = dev(gt_leaf)
    ref_q = _oracle.plonk_quotient_chunks_ex(host(wc), host(sc), host(zc), log_n, op, betas, gammas, alphas, gt_nat)
    got_q = host(ctx.plonk_quotient_chunks_ex(wl, sl, zl, log_n, rate_bits, gp, betas, gammas, alphas, d_gt))
    bad = np.argwhere(got_q != ref_q)
    assert bad.size == 0, "first mismatching (chunk, coefficient): %s" % bad[:4].tolist()
    assert (got_q != host(ctx.plonk_quotient_chunks(wl, sl, zl, log_n, rate_bits, gp, betas, gammas, alphas))).any()
    # the whole flow
    fp = fri(log_n, rate_bits=rate_bits, cap_height=cap_h, nq=5, arity=3, fpb=3)
    pis = [(11 * k + 3) % P for k in range(n_pi)]
    digest = (21, 22, 23, 24)
    ref = _oracle.plonk_prove_ex(wires, sig, log_n, op, fp, digest, pis, K)
    kw = dict(wires_oracle=w_or, wires_cap=w_cap, sigmas_oracle=s_or) if precommitted else {}
    got = ctx.plonk_prove_ex(d_w, d_s, log_n, gp, to_params(fp), digest, pis, gate_terms=d_gt, **kw)
    assert len(got) == len(ref) and int(got[0]) == 0x324b4c5050504953 and int(got[6]) == K and int(got[7]) == n_pi
    diff = np.nonzero(got != ref)[0]
    assert diff.size == 0, "first mismatch at word %d of %d" % (diff[0], len(ref))
    sig_cap = _oracle.Batch(sig, log_n, rate_bits=rate_bits, cap_height=cap_h).cap
    assert _oracle.plonk_verify_ex(got, sig_cap, op, fp, digest, K) == 0
    # a broken gate (one output cell) with honestly recomputed terms: the device proves, the verifier refuses at the quotient identity
    broken = wires.copy()
    broken[2, 5] = (int(broken[2, 5]) + 1) % P
    d_b = dev(broken)
    _, _, (bc, _, _) = ctx.commit_ex(d_b, log_n, rate_bits, cap_h)
    bt = np.zeros((K, m), dtype=np.uint64)
    bt[:, :nd] = _oracle.plonk_gate_terms_coset(host(bc), log_n, log_d, K)[:, rev]
    bad_pf = ctx.plonk_prove_ex(d_b, d_s, log_n, gp, to_params(fp), digest, pis, gate_terms=dev(bt))
    assert _oracle.plonk_verify_ex(bad_pf, sig_cap, op, fp, digest, K) == -210


def test_gate_term_argument_errors(ctx):
    import sipp_amd
    import torch
    gp = sipp_amd.PlonkParams(9, 2, 1)
    z = torch.zeros((9, 2048), dtype=torch.int64, device="cuda")
    zz = torch.zeros((2, 2048), dtype=torch.int64, device="cuda")
    L = ctx.L
    one = (C.c_uint64 * 1)(1)
    out = torch.zeros((2, 1024), dtype=torch.int64, device="cuda")
    # a count without a pointer, a pointer without a count
    assert L.sipp_plonk_quotient_chunks_ex(ctx.h, z.data_ptr(), z.data_ptr(), zz.data_ptr(), 10, 1, C.byref(gp), one, one, one, None, 3, out.data_ptr()) == -1
    assert L.sipp_plonk_quotient_chunks_ex(ctx.h, z.data_ptr(), z.data_ptr(), zz.data_ptr(), 10, 1, C.byref(gp), one, one, one, z.data_ptr(), 0, out.data_ptr()) == -1


# ---------------------------------------------------------------------------------------------------- gates as data (round 5)
@pytest.mark.parametrize("log_n,num_wires,num_routed,rate_bits,cap_h", [(10, 40, 24, 3, 2), (11, 136, 80, 3, 4), (10, 36, 16, 3, 0)])
def test_gates_as_data_proof_identical_to_oracle(ctx, log_n, num_wires, num_routed, rate_bits, cap_h):
    """sipp_plonk_prove_gates ("SIPPPLK3"): the outer flow with the circuit's gates evaluated ON THE DEVICE from the gate set's programs
    (arithmetic, base-sum, public-input and x^7 gates in two selector groups: tools/plonk_synth.py; the middle case has the column counts of
    CircuitConfig::standard_ecc_config, reference src/verifier_circuit.rs:213, and with them the RECURSION-SHAPED gate set of round 6:
    Poseidon, U32 multiply-add, random access, reducing, ... -- 118 gate constraints in three selector groups) -- constants_sigmas / wires / Z / quotient commitments,
    transcript, openings and FRI: the flat proof is the oracle's word for word and its verifier accepts it; with the constants_sigmas
    and wires oracles committed beforehand the same words come out; a malformed program is refused before any kernel runs."""
    import sipp_amd
    from tests.test_oracle_plonk import _synth
    ps, circ, wires, cs, gate, pis, pih = _synth(log_n, num_wires, num_routed, seed=40 + log_n)
    op, gp = _oracle.plonk_params(num_routed, 8, 2), sipp_amd.PlonkParams(num_routed, 8, 2)
    ofp = fri(log_n, rate_bits=rate_bits, cap_height=cap_h, nq=5, arity=4, fpb=3)
    gfp = to_params(ofp)
    digest = (11, 12, 13, 14)
    ref = _oracle.plonk_prove_gates(wires, cs, log_n, op, ofp, circ, digest, pis)
    gc = sipp_amd.PlonkCircuit.from_dict(circ)
    d_w, d_cs = dev(wires), dev(cs)
    got = ctx.plonk_prove_gates(d_w, d_cs, log_n, gp, gfp, gc, digest, pis)
    assert len(got) == len(ref)
    diff = np.nonzero(got != ref)[0]
    assert diff.size == 0, "first mismatch at word %d of %d" % (diff[0], len(ref))
    cs_cap = _oracle.Batch(cs, log_n, rate_bits=rate_bits, cap_height=cap_h).cap
    assert _oracle.plonk_verify_gates(got, cs_cap, op, ofp, circ, digest) == 0
    assert _verify.lib_plonk_verify(got, cs_cap, op, ofp, circ, digest) == 0          # the library's own verifier (sipp_plonk_verify_gates)
    # the caller's own commitments (constants_sigmas once per circuit; wires before the call): the same proof
    K = circ["num_constants"]
    cs_or, _cs_cap, keep1 = ctx.commit_ex(d_cs, log_n, rate_bits, cap_h)
    w_or, w_cap, keep2 = ctx.commit_ex(d_w, log_n, rate_bits, cap_h)
    assert cs_or.n_polys == K + num_routed and w_or.n_polys == num_wires and (_cs_cap == cs_cap).all()
    got2 = ctx.plonk_prove_gates(d_w, d_cs, log_n, gp, gfp, gc, digest, pis, wires_oracle=w_or, wires_cap=w_cap, cs_oracle=cs_or)
    assert (got2 == ref).all()
    del keep1, keep2
    # a broken witness is proved as it is and refused by the verifier (the quotient identity fails at zeta)
    w2 = wires.copy()
    w2[3, int(np.flatnonzero(gate == 1)[2])] ^= 1
    bad = ctx.plonk_prove_gates(dev(w2), d_cs, log_n, gp, gfp, gc, digest, pis)
    assert _oracle.plonk_verify_gates(bad, cs_cap, op, ofp, circ, digest) == -210 and _verify.lib_plonk_verify(bad, cs_cap, op, ofp, circ, digest) == 210
    # malformed circuits: an operand out of range, a program that runs past its words
    for mut in ("operand", "length"):
        c2 = dict(circ, programs=circ["programs"].copy())
        if mut == "operand":
            c2["programs"][4] = num_wires            # first monomial of arithmetic op 0: (kind, index) pairs start at word 3
        else:
            c2["programs"] = c2["programs"][:-3]
        with pytest.raises(sipp_amd.SippError) as e:
            ctx.plonk_prove_gates(d_w, d_cs, log_n, gp, gfp, sipp_amd.PlonkCircuit.from_dict(c2), digest, pis)
        assert e.value.code == -1


@pytest.mark.parametrize("log_n", [10, 14])
def test_witness_generators_match_the_oracle_and_feed_the_prover(ctx, log_n):
    """sipp_plonk_generate_witness (round 6: plonky2's generate_partial_witness for the recursion-shaped gate set, one lane per row): from
    the input cells alone -- the generated cells hold garbage -- the device rebuilds the wire table of oracle/plonk_witness.c (and of
    the numpy generator) bit for bit, family by family and all at once; the proof of the generated table is the oracle's proof of the
    numpy witness word for word; layouts that leave the table are SIPP_E_BADARG before any launch"""
    import sipp_amd
    from tests.test_oracle_plonk import _synth
    ps, circ, wires, cs, gate, pis, pih = _synth(log_n, 136, 80, seed=60 + log_n)
    K = circ["num_constants"]
    gens = ps.generators(circ)
    blank = ps.blank_generated(circ, wires, gate, value=0xDEADBEEF)
    ref = _oracle.plonk_generate_witness(blank, cs[:K], log_n, gens, pih)
    assert (ref == wires).all()
    d_cs = dev(cs)
    d_w = dev(blank)
    ctx.plonk_generate_witness(d_w, d_cs[:K], log_n, gens, pih)
    got = host(d_w)
    bad = np.argwhere(got != ref)
    assert bad.size == 0, "first mismatch: wire %d row %d (gate %d)" % (bad[0][0], bad[0][1], gate[bad[0][1]])
    # one family alone leaves every other row as it was
    for fam in (ps.GEN_POSEIDON, ps.GEN_U32_MUL_ADD, ps.GEN_REDUCING):
        d_w1 = dev(blank)
        g1 = [g for g in gens if g[0] == fam]
        ctx.plonk_generate_witness(d_w1, d_cs[:K], log_n, g1, pih)
        assert (host(d_w1) == _oracle.plonk_generate_witness(blank, cs[:K], log_n, g1, pih)).all()
    if log_n == 10:
        op, gp = _oracle.plonk_params(80, 8, 2), sipp_amd.PlonkParams(80, 8, 2)
        ofp = fri(log_n, rate_bits=3, cap_height=2, nq=5, arity=4, fpb=3)
        digest = (21, 22, 23, 24)
        want = _oracle.plonk_prove_gates(wires, cs, log_n, op, ofp, circ, digest, pis)
        pf = ctx.plonk_prove_gates(d_w, d_cs, log_n, gp, to_params(ofp), sipp_amd.PlonkCircuit.from_dict(circ), digest, pis)
        assert len(pf) == len(want) and (pf == want).all()
    for bad_gen in ((ps.GEN_POSEIDON, 2, 8, 0, 12, 40, 0, 0), (ps.GEN_ARITHMETIC, 0, 1, 35, 3, 4, 0, 0), (ps.GEN_CONSTANT, 0, 4, 2, 4, 0, 0, 0),
                    (ps.GEN_REDUCING, 5, 7, 40, 7, 0, 0, 0), (9, 0, 1, 0, 0, 0, 0, 0)):
        with pytest.raises(sipp_amd.SippError) as e:
            ctx.plonk_generate_witness(d_w, d_cs[:K], log_n, [bad_gen], pih)
        assert e.value.code == -1
    with pytest.raises(sipp_amd.SippError):                                  # a PublicInput generator needs the hash
        ctx.plonk_generate_witness(d_w, d_cs[:K], log_n, [g for g in gens if g[0] == ps.GEN_PUBLIC_INPUT], None)


@pytest.mark.parametrize("log_n,chain_len", [(10, 8), (14, 64)])
def test_levelled_witness_generation_matches_the_oracle(ctx, log_n, chain_len):
    """sipp_plonk_generate_witness_levels: the chained circuit's schedule (copy constraints from outputs to inputs of other rows) run level
    by level on the device -- two launches per level, captured as a hipGraph and replayed; the one-by-one route (SIPP_ROUTE_WITNESS_NO_GRAPH)
    and the replay give the oracle's table bit for bit, also after the inputs changed under the same graph; the proof of the generated
    table is the oracle's word for word; a schedule that leaves the table is SIPP_E_BADARG"""
    import sipp_amd
    from tests.test_oracle_plonk import _synth
    ps, circ, _w, _cs, _gate, pis, pih = _synth(log_n, 136, 80, seed=70 + log_n)
    wires, cs, gate = ps.witness(circ, log_n, 70 + log_n, pih, chain_len=chain_len)
    sc = ps.chain_schedule(log_n, chain_len)
    K, gens = circ["num_constants"], ps.generators(circ)
    blank = ps.blank_generated(circ, wires, gate, value=0xABCDEF, sched=sc)
    ref = _oracle.plonk_generate_witness_levels(blank, cs[:K], log_n, gens, pih, sc)
    assert (ref == wires).all()
    d_cs, d_w = dev(cs), dev(blank)
    sched = sipp_amd.PlonkSchedule.from_dict(sc)
    L = sipp_amd.lib()
    for route in (4, 0, 0):                                  # one by one, capture + replay, replay
        assert L.sipp_ctx_set_kernel_routes(ctx.h, route) == 0
        d_w.copy_(dev(blank))
        ctx.plonk_generate_witness_levels(d_w, d_cs[:K], log_n, gens, pih, sched)
        got = host(d_w)
        bad = np.argwhere(got != ref)
        assert bad.size == 0, "route %d: first mismatch wire %d row %d (gate %d)" % (route, bad[0][0], bad[0][1], gate[bad[0][1]])
    # other inputs under the SAME graph (same buffers, same schedule): the replay computes from what the table holds now
    blank2 = blank.copy()
    blank2[5, np.flatnonzero(gate == 8)] ^= np.uint64(1)            # a free input of every Poseidon row
    blank2[6, np.flatnonzero(gate == 1)] ^= np.uint64(2)            # an input of arithmetic op 1
    ref2 = _oracle.plonk_generate_witness_levels(blank2, cs[:K], log_n, gens, pih, sc)
    assert not (ref2 == ref).all()
    d_w.copy_(dev(blank2))
    ctx.plonk_generate_witness_levels(d_w, d_cs[:K], log_n, gens, pih, sched)
    assert (host(d_w) == ref2).all()
    if log_n == 10:
        op, gp = _oracle.plonk_params(80, 8, 2), sipp_amd.PlonkParams(80, 8, 2)
        ofp = fri(log_n, rate_bits=3, cap_height=2, nq=5, arity=4, fpb=3)
        digest = (31, 32, 33, 34)
        want = _oracle.plonk_prove_gates(wires, cs, log_n, op, ofp, circ, digest, pis)
        d_w.copy_(dev(blank))
        ctx.plonk_generate_witness_levels(d_w, d_cs[:K], log_n, gens, pih, sched)
        pf = ctx.plonk_prove_gates(d_w, d_cs, log_n, gp, to_params(ofp), sipp_amd.PlonkCircuit.from_dict(circ), digest, pis)
        assert len(pf) == len(want) and (pf == want).all()
        cs_cap = _oracle.Batch(cs, log_n, rate_bits=3, cap_height=2).cap
        assert _verify.lib_plonk_verify(pf, cs_cap, op, ofp, circ, digest) == 0
    for key, val in (("rows", 1 << log_n), ("copy_dst", 136 << log_n)):
        bad_sc = dict(sc, **{key: sc[key].copy()})
        bad_sc[key][3] = val
        with pytest.raises(sipp_amd.SippError) as e:
            ctx.plonk_generate_witness_levels(dev(blank), d_cs[:K], log_n, gens, pih, sipp_amd.PlonkSchedule.from_dict(bad_sc))
        assert e.value.code == -1
    assert L.sipp_ctx_set_kernel_routes(ctx.h, 0) == 0


@pytest.mark.parametrize("chain_len", [0, 16])
def test_circuit_data_build_prove_verify_over_host_arrays(chain_len):
    """sipp_circuit_build / _prove / _verify: `builder.build()`, `data.prove(pw)`, `data.verify(proof)` (reference src/verifier_circuit.rs:225,
    :253, :254) for a caller with host memory only.  From the wire table with its INPUT cells alone the proof is the oracle's proof of the
    full witness word for word -- row-local generators and the level schedule alike; the verifier data (cap, digest) equal the oracle's
    commitment / the caller's digest; the data's own verify accepts it and refuses a forged public input; a second proof with other
    public inputs under the same data verifies; a derived digest (NULL) is bound into the transcript; malformed builds are refused"""
    import sipp_amd
    from tests.test_oracle_plonk import _synth
    log_n, pis = 11, [7, 8, 9, 10, 11]
    ps, circ, _w, _cs, _gate, _pis, pih = _synth(log_n, 136, 80, seed=91, pis=pis)
    wires, cs, gate = ps.witness(circ, log_n, 91, pih, chain_len=chain_len)
    w_in = ps.witness(circ, log_n, 91, pih, inputs_only=True, chain_len=chain_len)[0]
    sc = ps.chain_schedule(log_n, chain_len) if chain_len else None
    op, gp = _oracle.plonk_params(80, 8, 2), sipp_amd.PlonkParams(80, 8, 2)
    ofp = fri(log_n, rate_bits=3, cap_height=4, nq=8, arity=4, fpb=4)
    gfp, gc = to_params(ofp), sipp_amd.PlonkCircuit.from_dict(circ)
    digest = (41, 42, 43, 44)
    L = sipp_amd.lib()
    ws = L.sipp_circuit_workspace_bytes(log_n, C.byref(gp), C.byref(gfp), C.byref(gc))
    assert ws > (4 << 30)
    c = sipp_amd.Ctx(workspace_bytes=ws)
    try:
        data = sipp_amd.CircuitData(c, log_n, gp, gfp, gc, cs, ps.generators(circ), sched=sc, digest=digest)
        cs_cap = _oracle.Batch(cs, log_n, rate_bits=3, cap_height=4).cap
        assert (data.cap == cs_cap).all() and [int(x) for x in data.digest] == list(digest)
        pf = data.prove(w_in, pis)
        want = _oracle.plonk_prove_gates(wires, cs, log_n, op, ofp, circ, digest, pis)
        assert len(pf) == len(want) and (pf == want).all()
        assert data.verify(pf) == (0, 0) and _oracle.plonk_verify_gates(pf, cs_cap, op, ofp, circ, digest) == 0
        forged = pf.copy()
        forged[-1] ^= 1
        assert data.verify(forged)[0] == -9
        # other public inputs under the same data: the PublicInput row follows the new hash
        pis2 = [1, 2, 3]
        pf2 = data.prove(w_in, pis2)
        assert data.verify(pf2) == (0, 0) and (pf2[-3:] == np.array(pis2, dtype=np.uint64)).all() and not (pf2[:64] == pf[:64]).all()
        data.close()
        # derived digest
        d2 = sipp_amd.CircuitData(c, log_n, gp, gfp, gc, cs, ps.generators(circ), sched=sc)
        assert d2.digest.any() and [int(x) for x in d2.digest] != list(digest)
        pf3 = d2.prove(w_in, pis)
        assert d2.verify(pf3) == (0, 0) and _oracle.plonk_verify_gates(pf3, cs_cap, op, ofp, circ, [int(x) for x in d2.digest]) == 0
        assert _oracle.plonk_verify_gates(pf3, cs_cap, op, ofp, circ, digest) != 0
        d2.close()
        # build / destroy gives every byte back (constants_sigmas, its oracle, the wire table, the schedule: ~17 MB at this size)
        from sipp_amd._lib import device_free_bytes
        free0 = device_free_bytes(0)
        for _ in range(4):
            d3 = sipp_amd.CircuitData(c, log_n, gp, gfp, gc, cs, ps.generators(circ), sched=sc)
            assert device_free_bytes(0) < free0 - (8 << 20)
            d3.close()
        assert abs(device_free_bytes(0) - free0) < (8 << 20)
        bad = dict(circ, programs=circ["programs"][:-3])
        with pytest.raises(sipp_amd.SippError):
            sipp_amd.CircuitData(c, log_n, gp, gfp, sipp_amd.PlonkCircuit.from_dict(bad), cs, ps.generators(circ), sched=sc)
        if sc:
            bad_sc = dict(sc, level_offsets=sc["level_offsets"][::-1].copy())
            with pytest.raises(sipp_amd.SippError):
                sipp_amd.CircuitData(c, log_n, gp, gfp, gc, cs, ps.generators(circ), sched=bad_sc)
    finally:
        c.close()


def test_witness_schedule_edge_cases(ctx):
    """the level schedule at its edges, device against oracle/plonk_witness.c: no generator at all (a no-op), ONE level without copies
    (= the row-local pass), empty levels in the middle (offsets that repeat), a schedule that names only SOME rows (the others keep
    their cells), the smallest table the entry point takes (2 rows)"""
    import sipp_amd
    from tests.test_oracle_plonk import _synth
    log_n = 10
    ps, circ, _w, _cs, _gate, pis, pih = _synth(log_n, 136, 80, seed=5)
    wires, cs, gate = ps.witness(circ, log_n, 5, pih, chain_len=4)
    sc = ps.chain_schedule(log_n, 4)
    K, gens, n = circ["num_constants"], ps.generators(circ), 1 << log_n
    blank = ps.blank_generated(circ, wires, gate, value=9, sched=sc)
    d_cs = dev(cs)

    def both(sched_dict, gs, table=blank):
        ref = _oracle.plonk_generate_witness_levels(table, cs[:K], log_n, gs, pih, sched_dict)
        d_w = dev(table)
        ctx.plonk_generate_witness_levels(d_w, d_cs[:K], log_n, gs, pih, sipp_amd.PlonkSchedule.from_dict(sched_dict))
        got = host(d_w)
        assert (got == ref).all()
        return got
    assert (both(sc, []) [:, np.flatnonzero(gate != 0)] == _oracle.plonk_generate_witness_levels(blank, cs[:K], log_n, [], pih, sc)[:, np.flatnonzero(gate != 0)]).all()
    one = {"n_levels": 1, "rows": np.arange(n, dtype=np.uint32), "level_offsets": np.array([0, n], dtype=np.uint32),
           "copy_src": np.zeros(0, dtype=np.uint64), "copy_dst": np.zeros(0, dtype=np.uint64), "copy_offsets": np.array([0, 0], dtype=np.uint32)}
    assert (both(one, gens) == _oracle.plonk_generate_witness(blank, cs[:K], log_n, gens, pih)).all()
    # empty levels: every level of the schedule twice, the second time with no rows and no copies
    lo, co = sc["level_offsets"], sc["copy_offsets"]
    gaps = dict(sc, n_levels=2 * sc["n_levels"], level_offsets=np.repeat(lo, 2)[1:].astype(np.uint32), copy_offsets=np.repeat(co, 2)[1:].astype(np.uint32))
    assert (both(gaps, gens) == wires).all()
    # only the Poseidon rows scheduled: everything else keeps its blanked cells
    rows8 = np.flatnonzero(gate == 8).astype(np.uint32)
    lv = sc["row_level"][rows8]
    o = np.argsort(lv, kind="stable")
    part = dict(sc, rows=rows8[o], level_offsets=np.searchsorted(lv[o], np.arange(sc["n_levels"] + 1)).astype(np.uint32))
    got = both(part, gens)
    assert (got[:, np.flatnonzero(gate == 2)] == blank[:, np.flatnonzero(gate == 2)]).all() and (got[12:24, rows8] != blank[12:24, rows8]).any()
    # two rows
    tiny_c, tiny_w = np.zeros((K, 2), dtype=np.uint64), np.arange(136 * 2, dtype=np.uint64).reshape(136, 2)
    tiny_c[0] = 1                                                               # two arithmetic rows
    tiny = {"n_levels": 1, "rows": np.array([0, 1], dtype=np.uint32), "level_offsets": np.array([0, 2], dtype=np.uint32),
            "copy_src": np.array([3 * 2 + 0], dtype=np.uint64), "copy_dst": np.array([0 * 2 + 1], dtype=np.uint64),
            "copy_offsets": np.array([0, 1], dtype=np.uint32)}
    ref = _oracle.plonk_generate_witness_levels(tiny_w, tiny_c, 1, gens[:1], pih, tiny)
    d_t = dev(tiny_w)
    ctx.plonk_generate_witness_levels(d_t, dev(tiny_c), 1, gens[:1], pih, sipp_amd.PlonkSchedule.from_dict(tiny))
    assert (host(d_t) == ref).all() and int(ref[0, 1]) == int(ref[3, 0])


def test_bench_outer_plonk_leg_runs_and_verifies():
    """bench.py's `outer_plonk` leg (plonky2 prove() at the standard_ecc_config column counts, gates as data) at a small size: the leg
    proves, the oracle's verifier accepts the proof, and the object carries its own roofline entries"""
    import bench
    r = bench.outer_plonk_leg(0, log_n=12, steps=1)
    assert r["verified"] is True and r["ms_per_proof"] > 0 and r["proof_words"] > 0
    assert r["shape"]["num_wires"] == 136 and r["shape"]["num_routed_wires"] == 80 and r["shape"]["rate_bits"] == 3
    assert r["roofline"]["transforms"]["algorithmic_bytes"] > 0 and r["roofline"]["leaf_hashing"]["permutations"] == (8 << 12) * (17 + 3 + 2)
    assert "plonk_quotient" in r["kernel_ms_per_proof"]
    # round 6: a recursion-shaped gate mix (>= 100 gate constraints, the Poseidon gate among them), the host column in the headline,
    # the quotient kernel against both bounds
    assert r["shape"]["num_gate_constraints"] >= 100 and "Poseidon" in r["shape"]["gates"] and "U32MulAdd" in r["shape"]["gates"]
    # the last step of round 6: witness generation on the device, inside the timed step, equal to the C port's table
    assert 0 < r["witness_generation_s"] < 0.05 and abs(r["end_to_end_s_per_proof"] - r["ms_per_proof"] * 1e-3) < 1e-9
    assert r["witness_matches_cpu_port"] is True and r["cpu_witness_generation_s"] > 0 and "witness_levels" in r["kernel_ms_per_proof"]
    assert abs(r["prove_below_witness_ms"] + r["witness_generation_ms"] - r["ms_per_proof"]) < 1e-6
    hh = r["host_to_host"]
    assert hh["same_proof"] is True and hh["verified"] is True and hh["ms_per_proof"] > 0 and hh["h2d_bytes_per_proof"] == 8 * 136 << 12
    w = r["witness_generation"]
    assert w["levels"] >= 60 and w["launches"] >= 100 and w["graph_replay_ms"] > 0 and w["launch_by_launch_ms"] > 0
    q = r["roofline"]["plonk_quotient"]
    assert q["bound"] == "valu" and 2000 < q["gate_products_per_point"] < 6000 and q["gate_products_per_point_as_written"] > 15000 and q["distinct_monomials"] < q["monomials"] / 3 and 0 < q["frac_valu_est"] < 1.5 and 0 < q["hbm"]["frac"] < 1
