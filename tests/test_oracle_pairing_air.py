"""The final-pairing AIR (API kind 6; reference src/bin/bls_aggregation.rs:76-77: `pairing_circuit(final_A, final_B)` connected to
`final_Z`) on the CPU: four readings of the value (the plain power of oracle/py/bn254.py, the operation schedule and its row program run
in big-int Python by tools/pairing_sched.py / tools/pairing_rows.py, the row program run by oracle/pairing.c), the trace against the
AIR program row by row, proofs through both verifiers, the mutation suite, and what is refused."""
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
import pairing_rows as PR  # noqa: E402


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


def test_row_program_shape():
    """one identity per row: every operation of the schedule is 12 component rows (24 for the inversion: witness, then check) plus a commit
    row, every point step 12 rows (slope witness, slope check, x3, y3, the two line coefficients), the Frobenius of Q 8"""
    from collections import Counter
    c = Counter(d[PR.F_TYP] for d in PR.ROWPROG)
    ops = Counter(r["fop"] for r in PS.SCHEDULE)
    assert PR.ROWS == 8192 and PR.N_ACTIVE == 7185 <= PR.ROWS
    assert c[PR.T_FMUL] == 12 * (ops[PS.F_MUL] + ops[PS.F_LINE] - 1) and c[PR.T_FCOPY] == 12 and c[PR.T_FFROB] == 12 * ops[PS.F_FROB]
    assert c[PR.T_FINVW] == c[PR.T_FINVC] == 12 and c[PR.T_FCOMMIT] == PS.N_ACTIVE - 1
    assert c[PR.T_GW] == c[PR.T_GSL] == c[PR.T_GX3] == c[PR.T_GY3] == c[PR.T_GL1] == c[PR.T_GL3] == 2 * 102 and c[PR.T_GFQ] == 8
    # a register is committed before it is read; the accumulator's twelve components are all produced before a commit
    loaded, filled = set(), set()
    for d in PR.ROWPROG:
        for k in (d[PR.F_RA], d[PR.F_RB]):
            assert k < 0 or k in loaded
        if d[PR.F_TYP] in (PR.T_FMUL, PR.T_FFROB, PR.T_FINVW, PR.T_FCOPY):
            filled.add(d[PR.F_T])
        if d[PR.F_TYP] == PR.T_FCOMMIT:
            assert filled == set(range(12))
            loaded.add(d[PR.F_LD])
            filled = set()


@pytest.mark.parametrize("seed", [1, 2])
def test_four_readings_of_the_value(seed):
    Pt, Q = points(seed)
    want = bn.pairing(Pt, Q)
    assert PS.simulate(Pt, Q) == want
    assert PR.simulate_rows(Pt, Q) == want
    assert (_oracle.pairing(record(Pt, Q)[:48]) == np.array(bn.f12_to_u32(want), dtype=np.uint32)).all()


def test_generator_pairing():
    assert (_oracle.pairing(record(bn.G1, bn.G2)[:48]) == np.array(bn.f12_to_u32(bn.pairing(bn.G1, bn.G2)), dtype=np.uint32)).all()


def _layout():
    import re
    names = "PX PY QX QY Q1X Q1Y Q2X Q2Y TX TY QSX QSY FXC FYC SR GC A B CACC REG RES".split()
    vals = [int(x) for x in re.search(r"AIR_PAIRING_LAYOUT_U8\[\d+\] = \{(.*?)\}", open("data/air_tables.h").read()).group(1).split(",")]
    return dict(zip(names, vals))


def test_trace_cells_follow_the_python_row_program_and_satisfy_the_air():
    """every primary cell of the C trace equals what tools/pairing_rows.py::simulate_rows holds on that row; every row satisfies every
    constraint of the AIR program (orc_trace_check_row)"""
    Pt, Q = points(5)
    tr = _oracle.Trace(6, record(Pt, Q).reshape(1, 144))
    assert tr.log_n == 14 and tr.num_io == 2 and tr.width == 2716
    t = tr.array()
    assert all(tr.check_row(r) == -1 for r in range(1 << tr.log_n))
    rows = []
    PR.simulate_rows(Pt, Q, rows)
    lay = _layout()

    def fq_at(col, row, checked=False):
        if not checked:
            return sum(int(t[col + i, row]) << (16 * i) for i in range(16))
        return sum((int(t[col + 2 * i, row]) + 256 * int(t[col + 2 * i + 1, row])) << (16 * i) for i in range(16))

    first = lambda typ: next(i for i, d in enumerate(PR.ROWPROG) if d[PR.F_TYP] == typ)
    probe = {0, 7, 8, 9, 19, 20, 25, 31, 32, 33, first(PR.T_FINVW), first(PR.T_FINVW) + 11, first(PR.T_FINVC) + 5, first(PR.T_FFROB) + 3,
             first(PR.T_FCOMMIT), first(PR.T_FCOMMIT) + 1, 4000, PR.N_ACTIVE - 1, PR.N_ACTIVE, 8191}
    for r in sorted(probe):
        v = rows[r]
        assert fq_at(lay["RES"], r, True) == v["res"], r
        assert [fq_at(lay["GC"] + 16 * i, r) for i in range(2)] == list(v["gc"])
        for i in range(12):
            assert fq_at(lay["A"] + 16 * i, r) == v["A"][i] and fq_at(lay["B"] + 16 * i, r) == v["B"][i] and fq_at(lay["CACC"] + 16 * i, r) == v["cacc"][i]
            for k in range(PR.NREG):
                assert fq_at(lay["REG"] + 192 * k + 16 * i, r) == v["regs"][k][i]
        for q in range(10):
            assert fq_at(lay["SR"] + 16 * q, r) == v["S"][q // 2][q % 2]
        assert [fq_at(lay[nm] + 16 * c, r) for nm in ("TX", "TY") for c in range(2)] == v["T"]
        assert [fq_at(lay[nm] + 16 * c, r) for nm in ("QSX", "QSY") for c in range(2)] == v["QS"]
        assert [fq_at(lay[nm] + 16 * c, r) for nm in ("Q1X", "Q1Y") for c in range(2)] == v["Q1"]
        assert [fq_at(lay[nm] + 16 * c, r) for nm in ("Q2X", "Q2Y") for c in range(2)] == v["Q2N"]
    # the second block is the padding copy of the record
    assert (t[1:, 8192:] == t[1:, :8192])[: lay["RES"] - 1].all()


def test_proofs_verify_with_both_readers_and_tampering_is_refused():
    cfg, pycfg = small_cfg()
    recs = np.stack([record(bn.G1, bn.G2), record(*points(7))])          # (three records, a padded block: tests/test_gpu_pairing_stark.py)
    pf = _oracle.stark_prove(6, recs, cfg)
    assert int(pf[1]) == 6 and int(pf[2]) == 14 and int(pf[3]) == 2 and int(pf[4]) == 2716 and int(pf[5]) == 252
    assert _oracle.stark_verify(pf, cfg) == 0
    assert sv.verify(pf, pycfg) is None
    # a public-input word of Z, of Q: both verifiers refuse
    for k, off in enumerate((len(pf) - 1, len(pf) - 144 + 20)):          # a word of the second record's Z, of its Q
        bad = pf.copy()
        bad[off] ^= 1
        assert _oracle.stark_verify(bad, cfg) != 0
        if k == 0:                                         # the Python reading (20 s per proof of this width) on one of the two
            assert sv.verify(bad, pycfg) is not None


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
# RES is the gadget's unknown; on rows whose identity does not contain it (flag `fres` = 0: a commit, a slope CHECK -- the slope sits in SR
# already --, the inversion's check, idle rows) it is range-checked and read by nothing.  A block's first row loads SR slot 0 from RES
# (the first component of pi(Q).x), so what that cell held before is never read.
FREE_CELLS = {("RES", rn) for rn in ("commit", "tangent_slope", "chord_slope", "inverse_check", "last_active", "idle", "block_last")} | {
    ("SR_slope", "next_block_first")}


def test_every_column_class_mutation_breaks_a_row_constraint_or_is_a_documented_free_cell():
    """one cell of every column class is changed on every ROW TYPE of the row program (a Frobenius-of-Q row, the first product's component
    rows, a commit, a slope witness / check, x3, y3, both line coefficients of a tangent and of a chord step, the copy of the first line,
    the inversion's witness and check rows, a Frobenius-map row, the last active row, an idle row, the block's last row and the next
    block's first): a constraint of that row or of the row before must fail -- except for the cells the AIR leaves FREE on purpose, listed
    below with the reason.  The list is exhaustive: anything else undetected fails the test."""
    lay = _layout()
    Pt, Q = points(31)
    tr = _oracle.Trace(6, record(Pt, Q).reshape(1, 144))
    arr = tr.array()
    a = tr.air
    first = lambda typ, sk=None, nth=0: [i for i, d in enumerate(PR.ROWPROG) if d[PR.F_TYP] == typ and (sk is None or d[PR.F_SK] == sk)][nth]
    rows = {
        "frob_of_q": 2, "frob_of_q_last": 7,
        "product": first(PR.T_FMUL, None, 5), "copy_line": first(PR.T_FCOPY, None, 3), "commit": first(PR.T_FCOMMIT),
        "slope_witness": first(PR.T_GW, PR.SK_TANGENT), "tangent_slope": first(PR.T_GSL, PR.SK_TANGENT, 1), "chord_slope": first(PR.T_GSL, PR.SK_CHORD),
        "x3_tangent": first(PR.T_GX3, PR.SK_TANGENT), "x3_chord": first(PR.T_GX3, PR.SK_CHORD, 1), "y3": first(PR.T_GY3, None, 1),
        "line1": first(PR.T_GL1), "line3": first(PR.T_GL3, None, 1), "step_end": first(PR.T_GL3, None, 1),
        "inverse_witness": first(PR.T_FINVW, None, 4), "inverse_check": first(PR.T_FINVC, None, 7), "frobenius_map": first(PR.T_FFROB, None, 9),
        "last_active": PR.N_ACTIVE - 1, "idle": PR.N_ACTIVE + 5, "block_last": PR.ROWS - 1, "next_block_first": PR.ROWS,
    }
    prog = np.ctypeslib.as_array(a.prog, shape=(a.prog_len,))
    q0 = int(prog[7 + 3])
    cols = {"PX": lay["PX"] + 3, "PY": lay["PY"] + 1, "QX": lay["QX"] + 17, "QY": lay["QY"] + 2, "Q1X": lay["Q1X"] + 4, "Q1Y": lay["Q1Y"] + 20,
            "Q2X": lay["Q2X"] + 1, "Q2Y": lay["Q2Y"] + 9, "Q2Y_c1": lay["Q2Y"] + 19, "TX": lay["TX"] + 5, "TY": lay["TY"] + 18, "QSX": lay["QSX"] + 2,
            "QSY": lay["QSY"] + 19, "FXC": lay["FXC"] + 7, "FYC": lay["FYC"] + 16, "GC": lay["GC"] + 16, "A": lay["A"] + 16 * 3 + 2, "B": lay["B"] + 16 * 6 + 1,
            "CACC": lay["CACC"] + 16 * 5 + 3, "REG_result": lay["REG"] + 192 * PR.RESULT_REG + 16 * 5 + 3, "REG_other": lay["REG"] + 192 * 3 + 7,
            "SR_slope": lay["SR"] + 5, "SR_x3": lay["SR"] + 32 + 3, "SR_y3": lay["SR"] + 64 + 17, "SR_l1": lay["SR"] + 96 + 8, "SR_l3": lay["SR"] + 128 + 20,
            "RES": lay["RES"] + 3, "sign": int(prog[1]), "q": q0 + 2, "carry": int(prog[2]) + 1, "carry_last": a.n_main - 1}
    free = set(FREE_CELLS)
    undetected, listed_but_detected = [], []
    for cname, col in cols.items():
        assert 0 < col < a.n_main, (cname, col)
        for rn, r in rows.items():
            assert tr.check_row(r) == -1 and tr.check_row(r - 1 if r else r) == -1, rn
            old = int(arr[col, r])
            arr[col, r] = old ^ 1
            seen = tr.check_row(r) != -1 or (r > 0 and tr.check_row(r - 1) != -1)
            arr[col, r] = old
            if not seen and cname == "sign" and not arr[q0:q0 + 34, r].any():
                continue                                   # the gadget's quotient is zero on this row: + 0 = - 0, the sign bit carries nothing
            if not seen and (cname, rn) not in free:
                undetected.append((cname, rn))
            if seen and (cname, rn) in free:
                listed_but_detected.append((cname, rn))
    assert not undetected, sorted(undetected)
    assert not listed_but_detected, listed_but_detected
