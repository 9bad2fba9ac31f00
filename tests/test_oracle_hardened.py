"""The hardened G1 / G2 exponentiation AIRs (API kinds 4 / 5) on the CPU oracle.

The plain chord rule  lam (Px - Rx) = Py - Ry  says nothing where the accumulator meets the running power (R = P: 0 lam = 0 for EVERY
lam), so for a crafted record (offset a small multiple of x) a cheating prover can walk on from a point of its choosing.  The test
below BUILDS that forgery through the oracle's test hook and shows both halves: the plain AIR's verifier accepts the false statement,
the hardened AIR has no witness for it -- and PROVES the true statement instead (R = P is a case of the variant: a flag eq, the sum
taken from the next row's double).  What the variant adds: x3 canonical (T3 = p - 1 - x3 with a borrow chain), eq / u / eqc, and on add
rows sum_j (Px_j - Rx_j) nz_j = u = bit (1 - eq)."""
import ctypes as C
import re
import os

import numpy as np
import pytest

from tests import _oracle
from oracle.py import bn254 as bn
from oracle.py import sipp_native as sn

ROOT = os.path.join(os.path.dirname(__file__), "..")


@pytest.fixture(scope="module")
def ios4():
    d = np.load("tests/golden/sipp_n4_ios.npz")
    return d["g1"], d["g2"]


def hard_layout(name):
    txt = open(os.path.join(ROOT, "data", "air_tables.h")).read()
    m = re.search(r"AIR_HARD_LAYOUT_%s\[13\] = \{([^}]*)\}" % name, txt)
    return [int(x) for x in m.group(1).split(",")][:6]    # nz, cb, T3, eq, u, eqc (then ng, inf, t1, v, w, NGV, cn)


def hard_layout_all(name):
    txt = open(os.path.join(ROOT, "data", "air_tables.h")).read()
    m = re.search(r"AIR_HARD_LAYOUT_%s\[13\] = \{([^}]*)\}" % name, txt)
    return dict(zip("nz cb T3 eq u eqc ng inf t1 v w NGV cn".split(), [int(x) for x in m.group(1).split(",")]))


@pytest.mark.parametrize("kind", [4, 5])
def test_rows_hold_and_the_plain_columns_are_unchanged(ios4, kind):
    ios = ios4[kind - 4]
    t, plain = _oracle.Trace(kind, ios), _oracle.Trace(kind - 4, ios)
    assert t.air.hardened == 1 and plain.air.hardened == 0 and t.air.kind == plain.air.kind == kind - 4
    n = 1 << t.log_n
    assert all(t.check_row(r) == -1 for r in range(0, n, 7))
    a, b = t.array(), plain.array()
    nz, cb, t3, eq, u, eqc = hard_layout("G%dH_U8" % (kind - 3))
    # the plain AIR's unchecked cells, then (shifted by the new unchecked columns) its checked cells, are the same values
    ub = plain.air.checked_base
    assert (a[:ub] == b[:ub]).all()
    shift = t.air.checked_base - ub
    assert (a[ub + shift: ub + shift + plain.air.n_checked] == b[ub: ub + plain.air.n_checked]).all()
    nc = 16 * (kind - 3)
    # the inequality witness sits exactly on the add rows whose bit is set
    bit = a[1 + 4 * nc]
    used = (a[nz: nz + nc] != 0).any(axis=0)
    assert (used == ((np.arange(n) % 2 == 0) & (bit == 1))).all()


@pytest.mark.parametrize("kind", [4, 5])
def test_mutations_of_the_new_cells_break_a_row(ios4, kind):
    t = _oracle.Trace(kind, ios4[kind - 4])
    arr = t.array()
    nz, cb, t3, eq, u, eqc = hard_layout("G%dH_U8" % (kind - 3))
    nc = 16 * (kind - 3)
    bit = arr[1 + 4 * nc]
    add_used = int(np.argwhere((np.arange(arr.shape[1]) % 2 == 0) & (bit == 1))[3][0])
    add_unused = int(np.argwhere((np.arange(arr.shape[1]) % 2 == 0) & (bit == 0))[3][0])
    j = int(np.argwhere(arr[nz: nz + nc, add_used] != 0)[0][0])
    for col, row in ((nz + j, add_used), (nz + (j + 1) % nc, add_used), (nz, add_unused), (cb + 3, 5), (cb + 14, 6), (t3 + 9, 5), (t3, 512),
                     (t3 + 2 * nc - 1, 7), (eq, add_used), (eq, add_unused), (u, add_used), (u, add_unused), (u, 7), (eqc, add_used + 1),
                     (eqc, 511)):
        old = int(arr[col, row])
        arr[col, row] = old ^ 1
        assert t.check_row(row) != -1, (col, row)
        arr[col, row] = old
        assert t.check_row(row) == -1
    # a non-canonical x3 (x3 + p: the same residue, other limbs) cannot be written: its T3 would need a final borrow
    x3 = t.air.checked_base + 2 * nc                      # lam | X3 | Y3, two cells per limb in the u8 variant
    row = 9
    limbs = [int(arr[x3 + 2 * i, row]) + 256 * int(arr[x3 + 2 * i + 1, row]) for i in range(16)]
    v = sum(l << (16 * i) for i, l in enumerate(limbs)) + bn.P
    if v < 1 << 256:
        for i in range(16):
            l = (v >> (16 * i)) & 0xFFFF
            arr[x3 + 2 * i, row], arr[x3 + 2 * i + 1, row] = l & 0xFF, l >> 8
        assert t.check_row(row) != -1


@pytest.mark.parametrize("kind", [4, 5])
def test_proof_verifies_and_is_bound_to_its_kind(ios4, kind):
    ios = ios4[kind - 4]
    pf = _oracle.stark_prove(kind, ios)
    assert int(pf[1]) == kind and _oracle.stark_verify(pf) == 0
    other = pf.copy()
    other[1] = kind - 4                                   # presented as a proof of the plain AIR: shapes do not match
    assert _oracle.stark_verify(other) != 0


def test_exceptional_addition_forgery_is_accepted_by_the_plain_air_and_the_hardened_one_proves_the_truth_instead():
    L = _oracle.load()
    L.orc_test_forge.argtypes = [C.c_int]
    x = bn.g1_mul(bn.G1, 77)
    # offset = x, odd exponent: the first addition is R + P with R = P.  The true output is 4 x; the forger claims whatever its walk ends in.
    rec = np.array([bn.g1_to_u32(x) + bn.g1_to_u32(x) + sn.exp_to_u32(3) + bn.g1_to_u32(bn.g1_mul(x, 4))] * 2, dtype=np.uint32)
    with pytest.raises(RuntimeError):
        _oracle.Trace(0, rec)                             # the honest generators refuse the record
    try:
        L.orc_test_forge(2 | 4)                           # outputs taken from the chain; a free slope where the chord rule is 0 = 0
        forged = _oracle.Trace(0, rec)
        assert all(forged.check_row(r) == -1 for r in range(0, 1024))
        pf = _oracle.stark_prove_trace(forged)
        assert _oracle.stark_verify(pf) == 0              # THE HOLE: a verified proof ...
        # ... for the second reading of the verifier too (a 3-query configuration: pure Python): the hole is the AIR's, not a verifier's
        from oracle.py import stark_verify as sv
        cfg3 = _oracle.default_config()
        cfg3.num_queries, cfg3.pow_bits = 3, 6
        assert sv.verify(_oracle.stark_prove_trace(forged, cfg3), dict(num_queries=3, pow_bits=6)) is None
        claimed = [int(v) for v in pf[-2 * 56:][40:56]]
        assert claimed != bn.g1_to_u32(bn.g1_mul(x, 4))   # ... of a false statement (out != offset + [3] x)
        # the hardened AIR: the forger (who claims the chord: eq = 0) finds no inequality witness; writing zeros breaks the add row
        hard = _oracle.Trace(4, rec)
        assert hard.check_row(0) != -1
        pfh = _oracle.stark_prove_trace(hard)
        assert _oracle.stark_verify(pfh) != 0
        assert sv.verify(_oracle.stark_prove_trace(hard, cfg3), dict(num_queries=3, pow_bits=6)) is not None
    finally:
        L.orc_test_forge(0)
    # ... while its honest generator PROVES the record: R = P is a case of the hardened AIR (eq = 1: the sum is the next row's double)
    good = _oracle.Trace(4, rec)
    assert all(good.check_row(r) == -1 for r in range(1024))
    nz, cb, t3, eq, u, eqc = hard_layout("G1H_U8")
    arr = good.array()
    # offset = x, e = 3: R = P = x, then R = P = 2 x (bit 1 set again), then R = P = 4 x with bit 2 clear: the flag three times, the
    # double handed over twice
    assert [int(arr[eq, r]) for r in range(8)] == [1, 0, 1, 0, 1, 0, 0, 0] and [int(arr[eqc, r]) for r in range(8)] == [0, 1, 0, 1, 0, 0, 0, 0]
    assert not arr[u, :8].any()
    pf = _oracle.stark_prove(4, rec)
    assert _oracle.stark_verify(pf) == 0
    assert [int(v) for v in pf[-2 * 56:][40:56]] == bn.g1_to_u32(bn.g1_mul(x, 4))                         # out = offset + [3] x = 4 x
    # the flag cannot be dropped (the chord has no witness) nor claimed where R != P
    arr[eq, 0], arr[u, 0] = 0, 1
    assert good.check_row(0) != -1
    arr[eq, 0], arr[u, 0] = 1, 0
    arr[eq, 6] = 1                                        # row 6: R = 4 x, P = 8 x
    assert good.check_row(6) != -1
    arr[eq, 6] = 0


def test_the_accumulator_passing_through_the_identity_is_proved_by_the_hardened_air_only():
    """offset = -x with an odd exponent: the first used addition is R + P with R = -P, the sum is the identity (flag ng, state bit inf,
    R keeps its cells), and the next used addition copies P: out = -x + [3] x = 2 x.  G1 and G2; the plain kinds refuse."""
    x1, x2 = bn.g1_mul(bn.G1, 77), bn.g2_mul(bn.G2, 99)
    g1 = np.array([bn.g1_to_u32(x1) + bn.g1_to_u32(bn.g1_neg(x1)) + sn.exp_to_u32(e) + bn.g1_to_u32(bn.g1_mul(x1, e - 1))
                   for e in (3, 0b1001, (1 << 200) | 1)], dtype=np.uint32)
    g2 = np.array([bn.g2_to_u32(x2) + bn.g2_to_u32(bn.g2_neg(x2)) + sn.exp_to_u32(e) + bn.g2_to_u32(bn.g2_mul(x2, e - 1))
                   for e in (3, 0b101)], dtype=np.uint32)
    for base, recs, name in ((0, g1, "G1H_U8"), (1, g2, "G2H_U8")):
        with pytest.raises(RuntimeError):
            _oracle.Trace(base, recs)
        t = _oracle.Trace(base + 4, recs)
        lay = hard_layout_all(name)
        arr = t.array()
        assert all(t.check_row(r) == -1 for r in range(1 << t.log_n))
        # row 0: ng and w; inf from row 1 until the next set bit's add row, where v copies P
        assert (int(arr[lay["ng"], 0]), int(arr[lay["w"], 0]), int(arr[lay["u"], 0]), int(arr[lay["inf"], 0]), int(arr[lay["inf"], 1])) == (1, 1, 0, 0, 1)
        first_set = 2 if base == 0 else 2          # e = 3: bit 1 -> add row 2
        assert int(arr[lay["v"], first_set]) == 1 and int(arr[lay["inf"], first_set]) == 1 and int(arr[lay["inf"], first_set + 1]) == 0
        pf = _oracle.stark_prove(base + 4, recs)
        assert _oracle.stark_verify(pf) == 0
        # the flags cannot be dropped or invented
        for col, row in ((lay["ng"], 0), (lay["inf"], 1), (lay["v"], first_set), (lay["w"], 0), (lay["t1"], 0), (lay["NGV"], 0), (lay["NGV"] + 3, 0),
                         (lay["ng"], 6), (lay["inf"], 9), (lay["cn"] + 2, 0)):
            old = int(arr[col, row])
            arr[col, row] = old ^ 1
            assert t.check_row(row) != -1 or t.check_row(row - 1) != -1, (col, row)
            arr[col, row] = old
    # what no affine record can say: the OUTPUT is the identity (offset = -x, e = 1)
    none = np.array([bn.g1_to_u32(x1) + bn.g1_to_u32(bn.g1_neg(x1)) + sn.exp_to_u32(1) + bn.g1_to_u32(x1)] * 2, dtype=np.uint32)
    for kind in (0, 4):
        with pytest.raises(RuntimeError):
            _oracle.Trace(kind, none)


def test_more_records_that_meet_the_running_power_are_proved_by_the_hardened_air_only():
    """offset = [2^i - (e mod 2^i)] x with bit i of e set: the accumulator equals the running power on add row i"""
    x = bn.g1_mul(bn.G1, 1234567)
    recs = []
    for e, i in ((0b1101, 2), (0b10001, 4), ((1 << 40) | 5, 40)):
        k = (1 << i) - (e % (1 << i))
        assert (e >> i) & 1
        recs.append(bn.g1_to_u32(x) + bn.g1_to_u32(bn.g1_mul(x, k)) + sn.exp_to_u32(e) + bn.g1_to_u32(bn.g1_mul(x, (k + e) % bn.R)))
    recs = np.array(recs, dtype=np.uint32)
    with pytest.raises(RuntimeError):
        _oracle.Trace(0, recs)
    t = _oracle.Trace(4, recs)
    nz, cb, t3, eq, u, eqc = hard_layout("G1H_U8")
    arr = t.array()
    for io, (e, i) in enumerate(((0b1101, 2), (0b10001, 4), ((1 << 40) | 5, 40))):
        assert int(arr[eq, 512 * io + 2 * i]) == 1 and int(arr[eqc, 512 * io + 2 * i + 1]) == 1
    assert all(t.check_row(r) == -1 for r in range(0, 1 << t.log_n, 1))
    assert _oracle.stark_verify(_oracle.stark_prove(4, recs)) == 0
