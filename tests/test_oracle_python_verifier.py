"""oracle/py/stark_verify.py -- the second reading of the STARK VERIFIER (public checks, Fiat-Shamir, every constraint of the AIR
program at zeta, lookups, permutation products, quotient identity, FRI) -- against proofs of the C oracle: it accepts what
oracle/stark.c accepts, for every AIR kind, and names the check a tampered proof fails."""
import numpy as np
import pytest

from tests import _oracle
from oracle.py import stark_verify as sv


def small_cfg(nq=3):
    cfg = _oracle.default_config()
    cfg.num_queries = nq
    cfg.pow_bits = 6
    return cfg, dict(num_queries=nq, pow_bits=6)


@pytest.fixture(scope="module")
def ios4():
    d = np.load("tests/golden/sipp_n4_ios.npz")
    return d["g1"], d["g2"], d["fq12"]


def mapg2_records():
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle", "py"))
    import bn254
    us = [(5, 7), (0, 3), (bn254.P - 2, 11)]
    return _oracle.map_to_g2(np.array([bn254.fq_to_u32(u[0]) + bn254.fq_to_u32(u[1]) for u in us], dtype=np.uint32))


def test_air_tables_parse_like_the_c_structs():
    periodic, p_limbs, airs, vper = sv.air_tables()
    assert len(periodic) == 12 and sum(l << (16 * i) for i, l in enumerate(p_limbs)) == sv.bn.P
    for a in airs:
        t = None
        kind = a["kind"] + 4 * a["hardened"]
        assert (a["n_vflag"], a["n_vconst"]) == ((len(vper["flagoff"]) - 1, 32) if kind == 6 else (0, 0))
        c = _oracle.load()
        c.orc_air_get.restype = __import__("ctypes").POINTER(_oracle.OrcAir)
        c.orc_air_get.argtypes = [__import__("ctypes").c_int, __import__("ctypes").c_uint]
        t = c.orc_air_get(kind, 16 if a["table_bits"] == 16 else 12).contents
        assert (t.n_main, t.checked_base, t.n_checked, t.n_aux, t.pi_per_io, t.prog_len, t.log_rows, t.hardened) == \
            (a["n_main"], a["checked_base"], a["n_checked"], a["n_aux"], a["pi_per_io"], len(a["prog"]), a["log_rows"], a["hardened"])


@pytest.mark.parametrize("kind", [0, 3, 5])      # G1 plain, MapToG2, G2 hardened (Fq12 below; G2 plain / G1 hardened share their code paths)
def test_python_verifier_accepts_c_proofs(ios4, kind):
    cfg, pycfg = small_cfg()
    ios = mapg2_records() if kind == 3 else ios4[kind % 4]
    pf = _oracle.stark_prove(kind, ios, cfg)
    assert _oracle.stark_verify(pf, cfg) == 0
    assert sv.verify(pf, pycfg) is None
    # wrong configuration: refused before any work
    assert sv.verify(pf) is not None


def test_python_verifier_accepts_the_fq12_proof_and_names_what_tampering_breaks(ios4):
    cfg, pycfg = small_cfg(3)
    pf = _oracle.stark_prove(2, ios4[2], cfg)
    assert sv.verify(pf, pycfg) is None
    W, P = int(pf[4]), int(pf[5])
    cap = 4 << 4
    op0 = 16 + 3 * cap                      # first opening word
    for off, what in ((op0 + 2 * 7, "quotient identity"),              # a trace opening at zeta
                      (op0 + 2 * (2 * W + 2 * P), "quotient identity"),  # a quotient chunk opening
                      (16 + 5, None),                                   # a word of the trace cap: every later challenge changes
                      (len(pf) - 10, None)):                            # a public-input word
        bad = pf.copy()
        bad[off] = (int(bad[off]) + 1) % _oracle.P
        why = sv.verify(bad, pycfg)
        assert why is not None and _oracle.stark_verify(bad, cfg) != 0
        if what:
            assert why.startswith(what), why
    # a sibling digest of the first query's trace path
    nf = 16 + 3 * cap + 2 * (2 * W + 2 * P + 4)
    arities = sv.g.reduction_arity_bits(4, 5, int(pf[2]), 1, 4)
    for r in range(len(arities)):
        nf += 4 * (1 << min(4, int(pf[2]) + 1 - sum(arities[:r + 1])))
    nf += 2 * int(pf[9]) + 1                # final polynomial, proof-of-work witness
    bad = pf.copy()
    bad[nf + W + 2] = (int(bad[nf + W + 2]) + 1) % _oracle.P
    assert sv.verify(bad, pycfg) == "Merkle path of oracle 0" and _oracle.stark_verify(bad, cfg) != 0


def test_default_configuration_84_queries(ios4):
    pf = _oracle.stark_prove(0, ios4[0])
    assert sv.verify(pf) is None
