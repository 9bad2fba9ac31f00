"""The library's verifiers of the GENERIC proofs (sipp_amd/csrc/verify.cpp: host C++, no GPU) against the oracle's, on the CPU:
`sipp_fri_verify_openings` (PolynomialBatch::verify_openings: any oracles / batches / FriParams, salted leaves) and
`sipp_plonk_verify_gates` (plonk/verifier.rs for the outer proof with the gate set as data: what `data.verify(proof)`, reference
src/verifier_circuit.rs:254, does with it).  Same verdict, same refusing stage on damaged proofs, same transcript afterwards."""
import random

import numpy as np
import pytest

from tests import _oracle
from tests._verify import lib_fri_verify, lib_plonk_verify
from tests.test_oracle_plonk import _synth, fri

P = _oracle.P


def test_outer_proofs_with_gates_as_data():
    ps, circ, wires, cs, gate, pis, pih = _synth(7)
    log_n = 7
    p = _oracle.plonk_params(circ["num_routed"], 8, 2)
    fp = fri(log_n, rate_bits=3, cap_height=2, nq=4, arity=2, fpb=3)
    digest = (9, 8, 7, 6)
    pf = _oracle.plonk_prove_gates(wires, cs, log_n, p, fp, circ, digest, pis)
    cs_cap = _oracle.Batch(cs, log_n, rate_bits=3, cap_height=2).cap
    both = lambda proof, c=circ, d=digest: (lib_plonk_verify(proof, cs_cap, p, fp, c, d), -_oracle.plonk_verify_gates(proof, cs_cap, p, fp, c, d))
    assert both(pf) == (0, 0)
    other = dict(circ, programs=circ["programs"].copy())
    other["programs"][1] = 2                       # another circuit's programs: the vanishing polynomial no longer meets the quotient
    assert both(pf, other) == (210, 210)
    assert both(pf, circ, (9, 8, 7, 5))[0] != 0 and both(pf, circ, (9, 8, 7, 5))[0] == both(pf, circ, (9, 8, 7, 5))[1]
    w2 = wires.copy()
    w2[12 + 5, int(np.flatnonzero(gate == 4)[3])] ^= 1       # a witness that breaks a gate: provable by nobody
    assert both(_oracle.plonk_prove_gates(w2, cs, log_n, p, fp, circ, digest, pis)) == (210, 210)
    rnd = random.Random(3)
    seen = set()
    for trial in range(160):
        bad = pf.copy()
        i = rnd.randrange(len(bad))
        how = trial % 4
        bad[i] = (int(bad[i]) ^ (1 << rnd.randrange(40))) if how == 0 else rnd.randrange(P) if how == 1 else P - 1 if how == 2 else P + rnd.randrange(99)
        if (bad == pf).all():
            continue
        a, b = both(bad)
        assert a == b and a != 0, (trial, i, a, b)
        seen.add(a)
    for cut in (1, 5, 300):
        a, b = both(pf[: len(pf) - cut])
        assert a == b != 0
    for word in (0, 1, 2, 5, 10, 11, 13):          # header fields: magic, log_n, routed wires, length, constraint count, public inputs, reserved
        bad = pf.copy()
        bad[word] = int(bad[word]) + 1
        a, b = both(bad)
        assert a == b and a in (201, 202, 203, 210), (word, a, b)         # (another log_n is a valid shape in which nothing fits: 210)
    assert 210 in seen and any(s in seen for s in (123, 124, 125, 126)), sorted(seen)


def test_recursion_shaped_proof():
    """the recursion-shaped gate set of bench.py's outer_plonk leg (Poseidon gate = the whole permutation, U32, random access, reducing ...)
    at 2^7 rows: accepted by both verifiers"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import plonk_synth as ps
    circ = ps.circuit_recursion_shaped()
    log_n = 7
    pis = [3, 1, 4, 1, 5]
    pih = [int(x) for x in _oracle.hash_no_pad(np.array(pis, dtype=np.uint64))]
    wires, cs, gate = ps.witness(circ, log_n, 11, pih)
    p = _oracle.plonk_params(circ["num_routed"], 8, 2)
    fp = fri(log_n, rate_bits=3, cap_height=2, nq=4, arity=2, fpb=3)
    pf = _oracle.plonk_prove_gates(wires, cs, log_n, p, fp, circ, (1, 2, 3, 4), pis)
    cs_cap = _oracle.Batch(cs, log_n, rate_bits=3, cap_height=2).cap
    assert _oracle.plonk_verify_gates(pf, cs_cap, p, fp, circ, (1, 2, 3, 4)) == 0
    assert lib_plonk_verify(pf, cs_cap, p, fp, circ, (1, 2, 3, 4)) == 0
    bad = pf.copy()
    bad[-2] ^= 1                 # a public input
    assert lib_plonk_verify(bad, cs_cap, p, fp, circ, (1, 2, 3, 4)) == -_oracle.plonk_verify_gates(bad, cs_cap, p, fp, circ, (1, 2, 3, 4)) != 0


@pytest.mark.parametrize("rate_bits,cap_height,arities,pow_rule,salted", [(1, 4, [4, 1], 0, False), (3, 2, [2, 3, 1], 1, False), (2, 0, [1, 1, 1, 1], 0, True)])
def test_opening_proofs_of_polynomial_batches(rate_bits, cap_height, arities, pow_rule, salted):
    log_n = 10
    n = 1 << log_n
    rng = np.random.default_rng(17 + rate_bits)
    a, b = _oracle.rand_field(rng, (5, n)), _oracle.rand_field(rng, (3, n))
    if salted:
        sa = _oracle.rand_field(rng, (4, n << rate_bits))
        oa = _oracle.SaltedBatch(a, log_n, rate_bits, cap_height, from_values=True, salt=sa)
    else:
        oa = _oracle.Batch(a, log_n, rate_bits=rate_bits, cap_height=cap_height)
    ob = _oracle.Batch(b, log_n, rate_bits=rate_bits, cap_height=cap_height, from_coeffs=True)
    fp = _oracle.fri_params(rate_bits=rate_bits, cap_height=cap_height, pow_bits=7, num_queries=9, pow_rule=pow_rule, hiding=1 if salted else 0,
                            arities=arities)
    z1, z2 = (int(rng.integers(1, 1 << 62)), int(rng.integers(1, 1 << 62))), (int(rng.integers(1, 1 << 62)), 0)
    batches = [(z1, [(0, 0, 5), (1, 1, 3)]), (z2, [(1, 0, 2), (0, 2, 4)])]
    pf = _oracle.fri_prove_openings([oa, ob], batches, log_n, fp, _oracle.challenger([7, 7, 7]))
    caps, ncols, n_salt = [oa.cap, ob.cap], [5, 3], [4 if salted else 0, 0]
    och = _oracle.challenger([7, 7, 7])
    want = _oracle.fri_verify_openings(pf, caps, ncols, n_salt, batches, log_n, fp, och)
    got, ch_after = lib_fri_verify(pf, caps, ncols, n_salt, batches, log_n, fp, _oracle.challenger([7, 7, 7]))
    assert want == 0 and got == 0
    assert ch_after == bytes(och)                        # the caller's transcript moved the same way
    rnd = random.Random(rate_bits)
    seen = set()
    for trial in range(80):
        bad = pf.copy()
        i = rnd.randrange(len(bad))
        bad[i] = (int(bad[i]) ^ (1 << rnd.randrange(40))) if trial % 3 == 0 else rnd.randrange(P) if trial % 3 == 1 else P + 5
        if (bad == pf).all():
            continue
        w = -_oracle.fri_verify_openings(bad, caps, ncols, n_salt, batches, log_n, fp, _oracle.challenger([7, 7, 7]))
        g, _ = lib_fri_verify(bad, caps, ncols, n_salt, batches, log_n, fp, _oracle.challenger([7, 7, 7]))
        assert g == w and g != 0, (trial, i, g, w)
        seen.add(g)
    assert len(seen) >= 4, sorted(seen)
    # another transcript in front: the challenges differ, nothing fits
    g, _ = lib_fri_verify(pf, caps, ncols, n_salt, batches, log_n, fp, _oracle.challenger([7, 7, 8]))
    assert g == -_oracle.fri_verify_openings(pf, caps, ncols, n_salt, batches, log_n, fp, _oracle.challenger([7, 7, 8])) != 0


def _fri_words(fp):
    return [fp.rate_bits, fp.cap_height, fp.pow_bits, fp.num_queries, fp.pow_rule, fp.hiding, fp.n_rounds] + [int(fp.arity_bits[i]) for i in range(32)]


def _run_fuzz_case(tmp_path, words, iters):
    import os
    import subprocess
    host = os.path.join(os.path.dirname(__file__), "host")
    subprocess.check_call(["make", "-C", host, "-s", "verify_fuzz_asan"])
    path = tmp_path / "case.bin"
    np.array([int(x) % (1 << 64) for x in words], dtype=np.uint64).tofile(path)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    out = subprocess.run([os.path.join(host, "verify_fuzz_asan"), "case", str(path), str(iters)], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0 and "verify fuzz ok" in out.stdout and " 0 accepted" in out.stdout, out.stdout + out.stderr


def test_generic_verifiers_on_damaged_proofs_under_the_sanitizers(tmp_path):
    """verify.cpp compiled with AddressSanitizer / UBSan (tests/host/verify_fuzz.cpp, `case` mode): an opening proof over a salted and a plain
    oracle, and an outer proof with its circuit, damaged 1,500 times each (header words, truncations with a consistent length word, random words,
    extensions): refusals only, nothing read out of bounds"""
    log_n, rate_bits, cap_height = 10, 2, 2
    n = 1 << log_n
    rng = np.random.default_rng(5)
    a, b = _oracle.rand_field(rng, (5, n)), _oracle.rand_field(rng, (3, n))
    oa = _oracle.SaltedBatch(a, log_n, rate_bits, cap_height, from_values=True, salt=_oracle.rand_field(rng, (4, n << rate_bits)))
    ob = _oracle.Batch(b, log_n, rate_bits=rate_bits, cap_height=cap_height)
    fp = _oracle.fri_params(rate_bits=rate_bits, cap_height=cap_height, pow_bits=5, num_queries=7, pow_rule=0, hiding=1, arities=[3, 2, 1])
    batches = [((123456789, 987654321), [(0, 0, 5), (1, 1, 3)]), ((55555, 0), [(1, 0, 2)])]
    pf = _oracle.fri_prove_openings([oa, ob], batches, log_n, fp, _oracle.challenger([1, 2]))
    ch = _oracle.challenger([1, 2])
    words = [1, log_n, 2, len(batches)] + _fri_words(fp)
    for o, nc, ns in ((oa, 5, 4), (ob, 3, 0)):
        words += [nc, ns] + [int(x) for x in np.asarray(o.cap).reshape(-1)]
    for pt, ranges in batches:
        words += [pt[0], pt[1], len(ranges)] + [x for r in ranges for x in r]
    words += [int(ch.state[i]) for i in range(12)] + [int(ch.in_buf[i]) for i in range(8)] + [int(ch.n_in)] + [int(ch.out_buf[i]) for i in range(8)] + [int(ch.n_out)]
    words += [len(pf)] + [int(x) for x in pf]
    _run_fuzz_case(tmp_path, words, 1500)
    ps, circ, wires, cs, gate, pis, pih = _synth(7)
    p = _oracle.plonk_params(circ["num_routed"], 8, 2)
    fp = fri(7, rate_bits=3, cap_height=2, nq=4, arity=2, fpb=3)
    digest = (9, 8, 7, 6)
    pf = _oracle.plonk_prove_gates(wires, cs, 7, p, fp, circ, digest, pis)
    cs_cap = _oracle.Batch(cs, 7, rate_bits=3, cap_height=2).cap
    words = [2, p.num_routed_wires, p.max_degree, p.num_challenges] + _fri_words(fp)
    words += [circ["num_wires"], circ["num_constants"], circ["num_selectors"], len(circ["gates"])] + [int(x) for g in circ["gates"] for x in g]
    words += [len(circ["programs"])] + [int(x) for x in circ["programs"]]
    words += list(digest) + [int(x) for x in np.asarray(cs_cap).reshape(-1)] + [len(pf)] + [int(x) for x in pf]
    _run_fuzz_case(tmp_path, words, 1500)
