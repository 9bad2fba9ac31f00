"""The final-pairing AIR (API kind 6; reference src/bin/bls_aggregation.rs:76-77: `pairing_circuit(final_A, final_B)` connected to
`final_Z`) on the CPU: three readings of the value (the plain power of oracle/py/bn254.py, the schedule run in big-int Python by
tools/pairing_sched.py, the schedule run by oracle/pairing.c), the trace against the AIR program row by row, proofs through both
verifiers, and what is refused."""
import os
import random
import sys

import numpy as np
import pytest

from tests import _oracle
from oracle.py import bn254 as bn
from oracle.py import stark_verify as sv

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
import pairing_sched as PS  # noqa: E402


def record(Pt, Q, Z=None):
    z = bn.pairing(Pt, Q) if Z is None else Z
    return np.array(bn.g1_to_u32(Pt) + bn.g2_to_u32(Q) + bn.f12_to_u32(z), dtype=np.uint32)


def points(seed):
    rnd = random.Random(seed)
    return bn.g1_mul(bn.G1, rnd.randrange(1, bn.R)), bn.g2_mul(bn.G2, rnd.randrange(1, bn.R))


def small_cfg(nq=2):
    cfg = _oracle.default_config()
    cfg.num_queries = nq
    cfg.pow_bits = 6
    return cfg, dict(num_queries=nq, pow_bits=6)


def test_the_pairing_is_arkworks_value_not_the_plain_reduced_one():
    """oracle/py/bn254.py: pairing = reduced ^ lambda, lambda = 2u(6u^2 + 3u + 1) (ark-ec's hard-part chain, recalled); still a
    non-degenerate bilinear map"""
    e = bn.pairing(bn.G1, bn.G2)
    assert e == bn.f12_pow(bn.pairing_reduced(bn.G1, bn.G2), bn.ARK_MULTIPLIER) and e != bn.pairing_reduced(bn.G1, bn.G2)
    assert bn.f12_pow(e, bn.R) == bn.F12_ONE and e != bn.F12_ONE
    assert bn.pairing(bn.g1_mul(bn.G1, 6), bn.g2_mul(bn.G2, 35)) == bn.f12_pow(e, 210)


def test_schedule_shape():
    assert PS.N_ACTIVE <= PS.ROWS == 512 and PS.NREG == 6
    ops = [r["fop"] for r in PS.SCHEDULE]
    assert ops.count(PS.F_INV) == 1 and ops.count(PS.F_LINE) == 64 + 36 + 2        # one inversion; a line per tangent / chord step
    assert [r["gop"] for r in PS.SCHEDULE].count(PS.G_TG) == 64
    # a register is never read before it was loaded, and loads happen only from rows that produce something
    loaded = set()
    for r in PS.SCHEDULE:
        for k in (r["ra"], r["rb"]):
            assert k < 0 or k in loaded
        if r["rd"] >= 0:
            loaded.add(r["rd"])


@pytest.mark.parametrize("seed", [1, 2])
def test_three_readings_of_the_value(seed):
    Pt, Q = points(seed)
    want = bn.pairing(Pt, Q)
    assert PS.simulate(Pt, Q) == want
    assert (_oracle.pairing(record(Pt, Q)[:48]) == np.array(bn.f12_to_u32(want), dtype=np.uint32)).all()


def test_generator_pairing():
    assert (_oracle.pairing(record(bn.G1, bn.G2)[:48]) == np.array(bn.f12_to_u32(bn.pairing(bn.G1, bn.G2)), dtype=np.uint32)).all()


def test_trace_cells_follow_the_python_schedule_and_satisfy_the_program():
    """every primary cell of the C trace equals what tools/pairing_sched.py::simulate computes for that row; every row satisfies every
    constraint of the AIR program (orc_trace_check_row)"""
    Pt, Q = points(5)
    tr = _oracle.Trace(6, record(Pt, Q).reshape(1, 144))
    assert tr.log_n == 10 and tr.num_io == 2
    t = tr.array()
    assert all(tr.check_row(r) == -1 for r in range(1 << tr.log_n))
    rows = []
    PS.simulate(Pt, Q, rows)
    lay = dict(zip("PX PY QX QY Q1X Q1Y Q2X Q2Y TX TY QSX QSY FXC FYC A B G REG C S0".split(),
                   [int(x) for x in __import__("re").search(r"AIR_PAIRING_LAYOUT_U8\[\d+\] = \{(.*?)\}", open("data/air_tables.h").read()).group(1).split(",")]))

    def fq_cells(col, row, checked):
        if not checked:
            return sum(int(t[col + i, row]) << (16 * i) for i in range(16))
        return sum((int(t[col + 2 * i, row]) + 256 * int(t[col + 2 * i + 1, row])) << (16 * i) for i in range(16))

    def f2_cells(col, row, checked=False):
        w = 32 if checked else 16
        return (fq_cells(col, row, checked), fq_cells(col + w, row, checked))

    for r in (0, 1, 2, 3, 100, 167, 168, 169, 170, 300, PS.N_ACTIVE - 1, PS.N_ACTIVE, 511):
        v = rows[r]
        assert f2_cells(lay["TX"], r) == v["T"][0] and f2_cells(lay["TY"], r) == v["T"][1]
        assert f2_cells(lay["QSX"], r) == v["QS"][0]
        for i in range(6):
            assert f2_cells(lay["A"] + 32 * i, r) == v["A"][i] and f2_cells(lay["B"] + 32 * i, r) == v["B"][i]
            assert f2_cells(lay["G"] + 32 * i, r) == v["G"][i]
            assert f2_cells(lay["C"] + 64 * i, r, True) == v["C"][i]
            for k in range(PS.NREG):
                assert f2_cells(lay["REG"] + 192 * k + 32 * i, r) == v["regs"][k][i]
        for sl in range(5):
            assert f2_cells(lay["S0"] + 64 * sl, r, True) == v["S"][sl]
    # the second block is the padding copy of the record
    assert (t[:, 512:1024][lay["A"]:lay["A"] + 192] == t[:, :512][lay["A"]:lay["A"] + 192]).all()


def test_proofs_verify_with_both_readers_and_tampering_is_refused():
    cfg, pycfg = small_cfg()
    recs = np.stack([record(bn.G1, bn.G2), record(*points(7))])          # (three records, a padded block: tests/test_gpu_pairing_stark.py)
    pf = _oracle.stark_prove(6, recs, cfg)
    assert int(pf[1]) == 6 and int(pf[2]) == 10 and int(pf[3]) == 2
    assert _oracle.stark_verify(pf, cfg) == 0
    assert sv.verify(pf, pycfg) is None
    # a public-input word of Z, of Q: both verifiers refuse
    for off in (len(pf) - 1, len(pf) - 144 + 20):          # a word of the second record's Z, of its Q
        bad = pf.copy()
        bad[off] ^= 1
        assert _oracle.stark_verify(bad, cfg) != 0 and sv.verify(bad, pycfg) is not None


def test_default_configuration_proof_verifies():
    pf = _oracle.stark_prove(6, record(*points(9)).reshape(1, 144))
    assert _oracle.stark_verify(pf) == 0


def test_wrong_result_is_not_provable():
    Pt, Q = points(3)
    wrong = bn.f12_mul(bn.pairing(Pt, Q), bn.pairing(bn.G1, bn.G2))          # another element of the target group
    with pytest.raises(RuntimeError, match="-8"):
        _oracle.stark_prove(6, record(Pt, Q, wrong).reshape(1, 144))


def test_forged_result_is_caught_by_the_verifier():
    """a prover that commits to the honest trace but publishes another Z (test hook bit 1 keeps the chain's output out of the record
    check -- here: the record check is skipped and the public inputs are overwritten) cannot satisfy the binding on the last row"""
    Pt, Q = points(4)
    tr = _oracle.Trace(6, record(Pt, Q).reshape(1, 144))
    tr.p.contents.pis[143] ^= 1                   # the statement's Z differs from the register the trace ends with
    cfg, pycfg = small_cfg()
    pf = _oracle.stark_prove_trace(tr, cfg)
    assert _oracle.stark_verify(pf, cfg) != 0 and sv.verify(pf, pycfg) is not None


def test_points_off_the_curve_or_outside_the_r_torsion_are_refused():
    Pt, Q = points(6)
    off = record((Pt[0], (Pt[1] + 1) % bn.P), Q, bn.F12_ONE)
    assert not _oracle.pairing_record_ok(off)
    with pytest.raises(RuntimeError):
        _oracle.stark_prove(6, off.reshape(1, 144))
    # a point of E'(Fp2) that is not in G2: on the twist, but of another order
    T = bn.g2_twist_point(12345)
    assert bn.g2_on_curve(T) and bn.g2_mul(T, bn.R) is not None
    outside = record(Pt, T, bn.F12_ONE)
    assert not _oracle.pairing_record_ok(outside)
    with pytest.raises(RuntimeError):
        _oracle.stark_prove(6, outside.reshape(1, 144))
    assert sv.records_ok(6, [int(x) for x in outside], 1, 144) == "Q outside the r-torsion"
    assert sv.records_ok(6, [int(x) for x in record(Pt, Q)], 1, 144) is None


# ---------------------------------------------------------------------------------------------------- AIR mutation suite
def _layout():
    import re
    names = "PX PY QX QY Q1X Q1Y Q2X Q2Y TX TY QSX QSY FXC FYC A B G REG C S0".split()
    vals = [int(x) for x in re.search(r"AIR_PAIRING_LAYOUT_U8\[\d+\] = \{(.*?)\}", open("data/air_tables.h").read()).group(1).split(",")]
    return dict(zip(names, vals))


def test_every_column_class_mutation_breaks_a_row_constraint_or_is_a_documented_free_cell():
    """one cell of every column class is changed on every ROW TYPE of the schedule (the Frobenius row 0, a squaring row, a tangent and a
    chord line row, the last chord, the inversion row, a Frobenius-map row, a plain product of the hard part, the last active row, an idle
    row, the block's last row and the next block's first): a constraint of that row or of the row before must fail -- except for the cells
    the AIR leaves FREE on purpose, listed below with the reason (they are never read: results of an idle unit, registers before their
    first load).  The list is exhaustive: anything else undetected fails the test."""
    lay = _layout()
    Pt, Q = points(31)
    tr = _oracle.Trace(6, record(Pt, Q).reshape(1, 144))
    arr = tr.array()
    a = tr.air
    S = PS.SCHEDULE
    first = lambda pred: next(i for i, r in enumerate(S) if pred(r))
    rows = {
        "frob_of_q": 0,
        "square": first(lambda r: r["fop"] == PS.F_MUL and r["gop"] == PS.G_IDLE),
        "tangent_line": first(lambda r: r["gop"] == PS.G_TG),
        "chord_line": first(lambda r: r["gop"] == PS.G_CH0),
        "last_chord": first(lambda r: r["gop"] == PS.G_CH2),
        "inverse": first(lambda r: r["fop"] == PS.F_INV),
        "frobenius_map": first(lambda r: r["fop"] == PS.F_FROB and PS.G_CONJ_COEF[r["gc"]]),
        "hard_part_product": PS.N_ACTIVE - 3,
        "last_active": PS.N_ACTIVE - 1,
        "idle": PS.N_ACTIVE + 5,
        "block_last": 511,
        "next_block_first": 512,
    }
    prog = np.ctypeslib.as_array(a.prog, shape=(a.prog_len,))
    cols = {"PX": lay["PX"] + 3, "PY": lay["PY"] + 1, "QX": lay["QX"] + 17, "QY": lay["QY"] + 2, "Q1X": lay["Q1X"] + 4, "Q1Y": lay["Q1Y"] + 20,
            "Q2X": lay["Q2X"] + 1, "Q2Y": lay["Q2Y"] + 9, "TX": lay["TX"] + 5, "TY": lay["TY"] + 18, "QSX": lay["QSX"] + 2, "QSY": lay["QSY"] + 19,
            "FXC": lay["FXC"] + 7, "FYC": lay["FYC"] + 16, "A": lay["A"] + 16 * 3 + 2, "B": lay["B"] + 16 * 6 + 1, "G": lay["G"] + 16 * 2,
            "REG_result": lay["REG"] + 192 * PS.RESULT_REG + 16 * 5 + 3, "REG_other": lay["REG"] + 192 * 3 + 7,
            "C": lay["C"] + 32 * 4 + 6, "S0": lay["S0"] + 5, "S1": lay["S0"] + 64 + 3, "S2": lay["S0"] + 128 + 40, "S3": lay["S0"] + 192 + 8,
            "S4": lay["S0"] + 256 + 33, "sign_fq12": int(prog[1]), "q_fq12": int(prog[7 + 3]) + 2, "carry_fq12": int(prog[2]) + 1,
            "carry_last_gadget": a.n_main - 1}
    q0 = int(prog[7 + 3])
    step_rows = {"tangent_line", "chord_line", "last_chord"}
    f_rows = {"square", "tangent_line", "chord_line", "last_chord", "frobenius_map", "hard_part_product", "last_active"}   # sC = 1: C is the result
    free = set()
    for rn in rows:
        # G2 unit idle: its five results are unconstrained (range-checked only) and nothing loads them
        if rn not in step_rows and rn not in ("frob_of_q", "next_block_first"):
            free |= {(c, rn) for c in ("S0", "S1", "S2", "S3", "S4")}
        # Fq12 unit idle or the inversion row's right-hand side: C is not "the result" (sC = 0) -- on the inversion row it IS constrained (A C = 1)
        if rn not in f_rows and rn != "inverse":
            free.add(("C", rn))
    free |= {("S4", "frob_of_q"), ("S4", "next_block_first")}                # row 0 computes four values (pi(Q), -pi^2(Q)); the fifth slot is idle
    # a block's FIRST row starts free: registers, T, pi(Q), -pi^2(Q) are loaded by the schedule before anything reads them
    for c in ("Q1X", "Q1Y", "Q2X", "Q2Y", "TX", "TY", "REG_result", "REG_other"):
        free |= {(c, "frob_of_q"), (c, "next_block_first")}
    # the last chord (with -pi^2(Q)) gives the last LINE; its point is not loaded into T, so x3 / y3 feed nothing -- but they are still
    # tied to the slope by their gadgets, hence detected: nothing to list.  REG_other = register 3 may hold a dead value:
    undetected = []
    for cname, col in cols.items():
        assert 0 < col < a.n_main, (cname, col)
        for rn, r in rows.items():
            assert tr.check_row(r) == -1 and tr.check_row(r - 1 if r else r) == -1, rn
            old = int(arr[col, r])
            arr[col, r] = old ^ 1
            seen = tr.check_row(r) != -1 or (r > 0 and tr.check_row(r - 1) != -1)
            arr[col, r] = old
            if not seen and cname == "sign_fq12" and not arr[q0:q0 + 34, r].any():
                continue                                   # the gadget's quotient is zero on this row: + 0 = - 0, the sign bit carries nothing
            if not seen and (cname, rn) not in free:
                undetected.append((cname, rn))
    assert not undetected, undetected
