"""sipp_stark_verify -- the library's OWN verifier (sipp_amd/csrc/verify.cpp: host C++, no GPU) -- against the oracle's
(oracle/stark.c), on the CPU.  What it stands for: starky's native `verify_stark_proof`, which the reference's proof generators call
right after `prove`, and the checks `data.verify(proof)` rests on (src/verifier_circuit.rs:254).

Two independent programs read the same proofs: every kind and configuration must get the same verdict, and a DAMAGED proof must be
refused at the same stage by both (the stage numbers of include/sipp_hip.h are the oracle's return codes, negated)."""
import os
import random
import sys

import numpy as np
import pytest

from tests import _oracle

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle", "py"))
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def configs(**kw):
    """the same configuration for the library (StarkConfig) and the oracle (OrcConfig)"""
    import sipp_amd
    cfg, ocfg = sipp_amd.default_config(), _oracle.default_config()
    for k, v in kw.items():
        setattr(cfg, k, v)
        setattr(ocfg, k, v)
    return cfg, ocfg


def both(proof, cfg=None, ocfg=None):
    import sipp_amd
    return sipp_amd.stark_verify(proof, cfg), -_oracle.stark_verify(proof, ocfg)


@pytest.fixture(scope="module")
def ios4():
    g = np.load(os.path.join(GOLD, "sipp_n4_ios.npz"))
    return {0: g["g1"], 1: g["g2"], 2: g["fq12"], 4: g["g1"], 5: g["g2"]}


@pytest.fixture(scope="module")
def proofs4(ios4):
    return {k: _oracle.stark_prove(k, v) for k, v in ios4.items()}


@pytest.mark.parametrize("kind", [0, 1, 2, 4, 5])
def test_accepts_the_oracles_proofs_of_every_exponentiation_kind(proofs4, kind):
    assert both(proofs4[kind]) == (0, 0)


def test_map_to_g2_and_pairing_proofs():
    import bn254 as bn
    import map_to_g2 as mg
    rnd = random.Random(5)
    msgs = [(rnd.randrange(bn.P), rnd.randrange(bn.P)) for _ in range(3)]
    recs = np.array([bn.fq_to_u32(u[0]) + bn.fq_to_u32(u[1]) + bn.g2_to_u32(mg.map_to_g2_without_cofactor_mul(u)) for u in msgs], dtype=np.uint32)
    pf = _oracle.stark_prove(3, recs)
    assert both(pf) == (0, 0)
    # the sign rule is a PUBLIC condition: the other root of the same x is on the curve, satisfies no constraint set the prover could not
    # fill -- and is refused by both verifiers before anything else is looked at
    bad = pf.copy()
    y = bn.u32_to_fq(list(recs[0, 32:40])), bn.u32_to_fq(list(recs[0, 40:48]))
    neg = bn.fq_to_u32((-y[0]) % bn.P) + bn.fq_to_u32((-y[1]) % bn.P)
    nio = int(pf[3])
    bad[len(bad) - nio * 48 + 32: len(bad) - nio * 48 + 48] = neg
    assert both(bad) == (109, 109)
    cfg, ocfg = configs(num_queries=3, pow_bits=5)
    Pt, Q = bn.g1_mul(bn.G1, 77), bn.g2_mul(bn.G2, 1234567)
    rec = np.array(bn.g1_to_u32(Pt) + bn.g2_to_u32(Q) + bn.f12_to_u32(bn.pairing(Pt, Q)), dtype=np.uint32).reshape(1, 144)
    pp = _oracle.stark_prove(6, rec, ocfg)
    assert both(pp, cfg, ocfg) == (0, 0)
    assert both(pp) == (102, 102)                     # another configuration than the proof's
    # Q outside the r-torsion is a public refusal (109): a point of the twist that is not in G2
    T = bn.g2_twist_point(11)
    assert bn.g2_on_curve(T) and bn.g2_mul(T, bn.R) is not None
    off = pp.copy()
    nio = int(pp[3])
    for blk in range(nio):
        base = len(off) - (nio - blk) * 144
        off[base + 16: base + 48] = bn.g2_to_u32(T)
    assert both(off, cfg, ocfg) == (109, 109)


@pytest.mark.parametrize("kw", [dict(fs_rule=1, lookup_rule=1, pow_rule=1), dict(rate_bits=2, arity_bits=3, final_poly_bits=4, num_queries=30),
                                dict(rate_bits=3, arity_bits=2, cap_height=2, num_queries=20, pow_bits=8), dict(lookup_rule=1), dict(fs_rule=1),
                                dict(cap_height=0, arity_bits=1, final_poly_bits=6, num_queries=12, pow_bits=0)])
def test_other_configurations(ios4, kw):
    cfg, ocfg = configs(**kw)
    pf = _oracle.stark_prove(0, ios4[0], ocfg)
    assert both(pf, cfg, ocfg) == (0, 0)
    # and a proof is a proof of ITS configuration only
    other, oother = configs(**dict(kw, num_queries=kw.get("num_queries", 84) - 1))
    a, b = both(pf, other, oother)
    assert a == b != 0


def sections(pf, cfg_cap=4):
    """word ranges of a flat proof: header, the three caps, the openings, the FRI part, the public inputs"""
    W, P, Q = int(pf[4]), int(pf[5]), int(pf[6])
    nio, ppi = int(pf[3]), int(pf[11])
    cap = 4 << cfg_cap
    a = 16
    out = {"header": (0, 16), "trace_cap": (a, a + cap), "z_cap": (a + cap, a + 2 * cap), "q_cap": (a + 2 * cap, a + 3 * cap)}
    o = a + 3 * cap
    n_open = 2 * (2 * W + 2 * P + Q)
    out["openings"] = (o, o + n_open)
    out["fri"] = (o + n_open, len(pf) - nio * ppi)
    out["public_inputs"] = (len(pf) - nio * ppi, len(pf))
    return out


@pytest.mark.parametrize("kind", [0, 5, 2])
def test_damaged_proofs_are_refused_at_the_same_stage_by_both(proofs4, kind):
    """single-word damage in every section (a flipped bit, a word replaced by a random field element, by p - 1, by a non-canonical
    value), truncation, extension, swapped neighbours: 150 damaged copies per kind; none accepted, every refusal at the oracle's stage"""
    pf = proofs4[kind]
    rnd = random.Random(1000 + kind)
    sec = sections(pf)
    P = (1 << 64) - (1 << 32) + 1
    seen = set()
    for name, (lo, hi) in sec.items():
        for trial in range(18):
            bad = pf.copy()
            i = rnd.randrange(lo, hi)
            how = trial % 4
            if how == 0:
                bad[i] = int(bad[i]) ^ (1 << rnd.randrange(0, 32 if name in ("header", "public_inputs") else 63))
            elif how == 1:
                bad[i] = rnd.randrange(P)
            elif how == 2:
                bad[i] = P - 1
            else:
                bad[i] = P + rnd.randrange(1 << 31)
            if (bad == pf).all():
                continue
            a, b = both(bad)
            assert a == b and a != 0, (name, i, how, a, b)
            seen.add(a)
    for cut in (1, 7, 100, len(pf) // 2):
        a, b = both(pf[: len(pf) - cut])
        assert a == b != 0
    a, b = both(np.concatenate([pf, np.zeros(3, dtype=np.uint64)]))
    assert a == b != 0
    lo, hi = sec["fri"]
    for _ in range(10):
        bad = pf.copy()
        i = rnd.randrange(lo, hi - 1)
        bad[i], bad[i + 1] = bad[i + 1], bad[i]
        if (bad == pf).all():
            continue
        a, b = both(bad)
        assert a == b != 0
        seen.add(a)
    # the damage reached the late stages too (constraints at zeta, Merkle paths, folds), not only the header checks
    assert {110, 123} <= seen or {110, 124} <= seen or {110, 125} <= seen, sorted(seen)
    assert any(s >= 130 for s in seen), sorted(seen)


def test_a_witness_that_breaks_a_constraint_is_refused_at_the_quotient(ios4):
    """the prover commits to whatever the trace holds: one changed cell of a gadget's quotient -> both verifiers stop at stage 110"""
    tr = _oracle.Trace(0, ios4[0])
    arr = tr.array()
    a = tr.air
    prog = np.ctypeslib.as_array(a.prog, shape=(a.prog_len,))
    qcol = int(prog[7 + 3])
    arr[qcol, 5] = (int(arr[qcol, 5]) + 1) % (1 << 8 if a.cells_per_limb == 2 else 1 << 16)
    pf = _oracle.stark_prove_trace(tr)
    assert both(pf) == (110, 110)


def test_bad_arguments():
    import sipp_amd
    L = sipp_amd.lib()
    assert L.sipp_stark_verify(None, 0, None, None) == -1
    cfg = sipp_amd.default_config()
    cfg.num_challenges = 3
    with pytest.raises(sipp_amd.SippError):
        sipp_amd.stark_verify(np.zeros(32, dtype=np.uint64), cfg)
    assert sipp_amd.stark_verify(np.zeros(8, dtype=np.uint64)) == 100


def test_damaged_proofs_under_the_sanitizers(tmp_path, ios4):
    """verify.cpp + the host permutation compiled INTO a test binary with AddressSanitizer / UBSan (tests/host/verify_fuzz.cpp): header
    fields at extreme values, truncations with a consistent length word, random words, shifted blocks -- every damaged proof ends in a
    refusal with a stage of the documented range, nothing is read out of bounds (scripts/run_asan.sh runs 20,000 per kind)"""
    import subprocess
    host = os.path.join(os.path.dirname(__file__), "host")
    subprocess.check_call(["make", "-C", host, "-s", "verify_fuzz_asan"])
    path = tmp_path / "proof.bin"
    _oracle.stark_prove(1, ios4[1]).tofile(path)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    out = subprocess.run([os.path.join(host, "verify_fuzz_asan"), str(path), "400", "3"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0 and "verify fuzz ok" in out.stdout and " 0 accepted" in out.stdout, out.stdout + out.stderr
