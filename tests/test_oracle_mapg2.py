"""MapToG2 on the CPU (reference src/bin/bls_aggregation.rs:65, :100-104): the two readings of the
map agree value for value, the derived constants agree three ways, every row of the trace satisfies the AIR, a proof verifies, and
the AIR / the public checks refuse what they must (PARITY UNPINNED: these tests are what stands behind the specification)."""
import os
import random
import re
import sys

import numpy as np
import pytest

from tests import _oracle

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle", "py"))
import bn254  # noqa: E402
import map_to_g2 as M  # noqa: E402

ROOT = os.path.join(os.path.dirname(__file__), "..")


def words_of(us):
    return np.array([bn254.fq_to_u32(u[0]) + bn254.fq_to_u32(u[1]) for u in us], dtype=np.uint32)


def messages(n, seed):
    rnd = random.Random(seed)
    return [(rnd.randrange(bn254.P), rnd.randrange(bn254.P)) for _ in range(n)]


def test_svdw_parameters():
    """Z = 1 meets RFC 9380's conditions for y^2 = x^3 + 3/(9+u) (appendix H.1 find_z_svdw), c3 has sgn0 = 0, and 9 + u is a
    non-residue (what the AIR's "not a square" witnesses rest on)"""
    gz = M.g(M.Z)
    h = bn254.f2_mul(bn254.f2_neg(bn254.f2_scal(bn254.f2_mul(M.Z, M.Z), 3)), bn254.f2_inv(bn254.f2_scal(gz, 4)))
    assert gz != (0, 0) and h != (0, 0) and M.is_square(h)
    assert M.is_square(gz) or M.is_square(M.g(bn254.f2_scal(M.Z, (bn254.P - 1) * bn254.inv(2) % bn254.P)))
    assert M.sgn0(M.C3) == 0 and bn254.f2_mul(M.C3, M.C3) == bn254.f2_neg(bn254.f2_mul(M.C1, (3, 0)))
    assert not M.is_square(bn254.XI)
    t = 6 * bn254.U ** 2 + 1                       # trace of Frobenius: r = p + 1 - t, #E'(Fp2) = r (p - 1 + t)
    assert bn254.R == bn254.P + 1 - t and bn254.G2_COFACTOR == bn254.P - 1 + t


def test_constants_agree_between_the_generated_header_and_python():
    txt = open(os.path.join(ROOT, "sipp_amd", "csrc", "mapg2_constants.h")).read()
    def arr(name):
        m = re.search(r"%s\[\d+\] = \{([^}]*)\}" % name, txt)
        return [int(x.strip().rstrip("u"), 16) for x in m.group(1).split(",")]
    for name, v in (("SIPP_MAPG2_C1", M.C1), ("SIPP_MAPG2_C2", M.C2), ("SIPP_MAPG2_C3", M.C3), ("SIPP_MAPG2_C4", M.C4),
                    ("SIPP_MAPG2_B", bn254.B2)):
        assert arr(name) == bn254.fq_to_u32(v[0]) + bn254.fq_to_u32(v[1]), name
    assert arr("SIPP_G2_GEN") == bn254.g2_to_u32(bn254.G2) and arr("SIPP_G2_GEN_NEG") == bn254.g2_to_u32(bn254.g2_neg(bn254.G2))
    h = sum(w << (32 * i) for i, w in enumerate(arr("SIPP_G2_COFACTOR")))
    assert h == bn254.G2_COFACTOR == 2 * bn254.P - bn254.R


def test_c_and_python_readings_of_the_map_agree():
    us = messages(40, 1) + [(0, 0), (0, 9), (3, 0), (bn254.P - 1, bn254.P - 1)]
    recs = _oracle.map_to_g2(words_of(us))
    seen = set()
    for u, r in zip(us, recs):
        w = M.witness(u)
        seen.add((w["e1"], w["e2"]))
        q = (w["XS"], w["Y"])
        assert list(r[16:]) == bn254.g2_to_u32(q)
        assert bn254.g2_on_curve(q) and M.sgn0(q[1]) == M.sgn0(u)
    assert seen == {(1, 0), (0, 1), (0, 0)}
    # cleared of the cofactor the image lies in G2
    q = M.map_to_g2(us[0])
    assert bn254.g2_on_curve(q) and bn254.g2_mul(q, bn254.R) is None


def air_tables():
    """the MapToG2 schedule and column layout as tools/air_gen.py emitted them (data/air_tables.h)"""
    txt = open(os.path.join(ROOT, "data", "air_tables.h")).read()
    def ints(name):
        m = re.search(r"%s(?:\[\d+\])+ = \{(.*?)\};" % name, txt, re.S)
        return [int(x) for x in re.findall(r"-?\d+", m.group(1))]
    slot = np.array(ints("AIR_MAPG2_SLOT_WIT")).reshape(8, 3)
    reg = np.array(ints("AIR_MAPG2_REG_WIT")).reshape(8, 6)
    lay = dict(zip("U ONE C1 C2 C3 C4 BB e1 e2 M1 M2 XS GXS REG RES RX1 RX2 RX3 RG1 RG2 RG3 z ZV TINV".split(), ints("AIR_MAPG2_LAYOUT_U8")))
    return slot, reg, lay


NAMES = ["T1", "TV1", "W", "TV3", "A4", "B4", "X2", "X1", "S1", "GX1", "S2", "GX2", "D", "E", "F", "X3", "S3", "GX3", "N1", "N2", "Y"]


@pytest.fixture(scope="module")
def trace():
    us = messages(9, 2)
    recs = _oracle.map_to_g2(words_of(us))
    return us, recs, _oracle.Trace(3, recs)


def test_every_row_satisfies_the_air_and_the_cells_are_the_python_witness(trace):
    us, recs, t = trace
    slot, reg, lay = air_tables()
    assert (t.log_n, t.num_io, t.air.log_rows, t.air.table_bits, t.air.pi_per_io) == (10, 128, 3, 8, 48)
    assert all(t.check_row(r) == -1 for r in range(1 << t.log_n))
    arr = t.array()
    assert lay["RES"] == t.air.checked_base
    # every witness sits in exactly one slot of its block, and is in a register wherever the schedule says so
    assert sorted(int(x) for x in slot.ravel() if x >= 0) == list(range(21))
    def fp2_u16(col, row):
        return tuple(sum(int(arr[col + 16 * c + l, row]) << (16 * l) for l in range(16)) for c in range(2))
    def fp2_chk(col, row):
        return tuple(sum((int(arr[col + 32 * c + 2 * l, row]) + 256 * int(arr[col + 32 * c + 2 * l + 1, row])) << (16 * l) for l in range(16))
                     for c in range(2))
    for io, u in enumerate(us):
        w = M.witness(u)
        for tt in range(8):
            row = 8 * io + tt
            assert (int(arr[lay["e1"], row]), int(arr[lay["e2"], row])) == (w["e1"], w["e2"])
            assert fp2_u16(lay["U"], row) == u
            for sl in range(3):
                if slot[tt, sl] >= 0:
                    assert fp2_chk(lay["RES"] + 64 * sl, row) == w[NAMES[slot[tt, sl]]], (tt, sl)
            for k in range(6):
                if reg[tt, k] >= 0:
                    assert fp2_u16(lay["REG"] + 32 * k, row) == w[NAMES[reg[tt, k]]], (tt, k)
        assert fp2_u16(lay["XS"], 8 * io + 7) == w["XS"] and fp2_chk(lay["RES"], 8 * io + 7) == w["Y"]
    # padding blocks repeat the last record
    assert (arr[1:t.air.n_main, 8 * 8:8 * 9] == arr[1:t.air.n_main, 1016:1024]).all()     # column 0 is the range table


def test_mutations_break_a_row_constraint(trace):
    """one cell of every column class changed on every row type: the constraints of that row or of the row before it (register and
    block-constant transitions look at the next row) must fail"""
    _, _, t = trace
    slot, reg, lay = air_tables()
    arr = t.array()
    a = t.air
    prog = np.ctypeslib.as_array(a.prog, shape=(a.prog_len,))
    sign0, carry0, q0 = int(prog[1]), int(prog[2]), int(prog[7 + 3])
    cols = {"U": lay["U"] + 3, "ONE": lay["ONE"], "C3": lay["C3"] + 17, "BB": lay["BB"] + 2, "M1": lay["M1"] + 1, "M2": lay["M2"],
            "XS": lay["XS"] + 4, "GXS": lay["GXS"] + 20, "ZV": lay["ZV"] + 1, "sign": sign0, "q": q0 + 3, "carry": carry0 + 1, "carry_last": a.n_main - 1}
    for sl in range(3):
        cols["RES%d" % sl] = lay["RES"] + 64 * sl + 7 + 32 * (sl & 1)
    for k in range(6):
        cols["REG%d" % k] = lay["REG"] + 32 * k + 5 + 16 * (k & 1)
    free = []
    for name, col in cols.items():
        for tt in range(8):
            row = 8 * 2 + tt
            old = int(arr[col, row])
            arr[col, row] = old ^ 1
            caught = t.check_row(row) != -1 or t.check_row(row - 1) != -1
            arr[col, row] = old
            assert t.check_row(row) == -1 and t.check_row(row - 1) == -1
            if not caught:
                free.append((name, tt))
    # what may be free: a register before its first load of the block and while nothing reads it again, an unused result slot --
    # none of them is read by a constraint that matters; every cell the schedule USES must be caught
    for name, tt in free:
        if name == "sign":      # the sign of a quotient that is zero (E = 0 exactly, e.g. x1 + x2 = 2 c2 without a wrap) is meaningless
            assert not arr[q0:q0 + 34, 8 * 2 + tt].any(), tt
        elif name.startswith("RES"):
            assert slot[tt, int(name[3:])] < 0, (name, tt)      # a slot without an identity on this row type
        elif name.startswith("REG"):
            assert reg[tt, int(name[3:])] < 0, (name, tt)       # a register before its first load of the block
        else:
            raise AssertionError("free cell %s on row type %d" % (name, tt))
    # the branch bits: claiming "square" for a non-square / "non-square" for a square breaks the block (e1 is constant over it)
    for io in range(9):
        e1 = int(arr[lay["e1"], 8 * io])
        arr[lay["e1"], 8 * io: 8 * io + 8] = 1 - e1
        assert any(t.check_row(8 * io + tt) != -1 for tt in range(8))
        arr[lay["e1"], 8 * io: 8 * io + 8] = e1
    # e2 matters only where e1 = 0
    tested = 0
    for io in range(9):
        if int(arr[lay["e1"], 8 * io]):
            continue
        tested += 1
        e2 = int(arr[lay["e2"], 8 * io])
        arr[lay["e2"], 8 * io: 8 * io + 8] = 1 - e2
        assert any(t.check_row(8 * io + tt) != -1 for tt in range(8))
        arr[lay["e2"], 8 * io: 8 * io + 8] = e2
    assert tested >= 1
    # a bit changed on ONE row of a block breaks the block-constant transition
    arr[lay["e1"], 8 * 3 + 4] ^= 1
    assert t.check_row(8 * 3 + 3) != -1 or t.check_row(8 * 3 + 4) != -1
    arr[lay["e1"], 8 * 3 + 4] ^= 1


def test_proof_verifies_and_public_checks_refuse_the_other_root():
    us = messages(3, 4)
    recs = _oracle.map_to_g2(words_of(us))
    proof = _oracle.stark_prove(3, recs)
    assert _oracle.stark_verify(proof) == 0
    # header: kind 3, 2^10 rows, 128 records (8 rows each)
    assert (int(proof[1]), int(proof[2]), int(proof[3])) == (3, 10, 128)
    # a record that claims -y is refused by the generator ...
    bad = recs.copy()
    y = (bn254.u32_to_fq(list(bad[1, 32:40])), bn254.u32_to_fq(list(bad[1, 40:48])))
    bad[1, 32:] = bn254.fq_to_u32((-y[0]) % bn254.P) + bn254.fq_to_u32((-y[1]) % bn254.P)
    with pytest.raises(RuntimeError):
        _oracle.Trace(3, bad)
    # ... and a proof whose public inputs say so is refused by the verifier's public check (-109), whatever its body
    forged = proof.copy()
    n_pi = 128 * 48
    forged[len(forged) - n_pi + 48 + 32: len(forged) - n_pi + 2 * 48] = bad[1, 32:]
    assert _oracle.stark_verify(forged) == -109
    forged = proof.copy()
    forged[len(forged) - n_pi + 7] = 0xFFFFFFFF       # u.c0 >= p
    assert _oracle.stark_verify(forged) == -108
    # any other tampering of the public inputs changes Fiat-Shamir: some later check fails
    forged = proof.copy()
    forged[len(forged) - n_pi + 16] ^= 1
    assert _oracle.stark_verify(forged) != 0


def inv0_messages():
    """the four u with u^2 g(Z) = +-1: the product the map inverts is zero there (inv0(0) = 0: x1 = x2 = -Z/2, x3 = Z)"""
    out = []
    for t in (bn254.f2_inv(M.C1), bn254.f2_neg(bn254.f2_inv(M.C1))):
        r = M.sqrt_even(t)
        if r is not None:
            out += [r, bn254.f2_neg(r)]
    return out


def test_the_inversion_of_zero_follows_inv0_and_is_provable():
    us = inv0_messages()
    assert len(us) >= 2
    recs = _oracle.map_to_g2(words_of(us))
    slot, reg, lay = air_tables()
    for u, r in zip(us, recs):
        w = M.witness(u)
        assert w["z"] == 1 and w["TV3"] == (0, 0) and w["X1"] == w["X2"] == M.C2
        q = (w["XS"], w["Y"])
        assert list(r[16:]) == bn254.g2_to_u32(q) and bn254.g2_on_curve(q)
    t = _oracle.Trace(3, recs)
    assert all(t.check_row(r) == -1 for r in range(8 * len(us) + 8))
    arr = t.array()
    z = arr[lay["z"], :8 * len(us)].reshape(len(us), 8)
    assert (z[:, 1] == 1).all() and z.sum() == len(us)          # the flag sits on the row that inverts, nowhere else
    # the flag cannot be claimed for a product that is not zero, nor dropped where it is
    other = _oracle.Trace(3, _oracle.map_to_g2(words_of(messages(2, 8))))
    oarr = other.array()
    oarr[lay["z"], 1] = 1
    oarr[lay["ZV"], 1] = 1
    assert other.check_row(1) != -1
    arr[lay["z"], 1] = 0
    arr[lay["ZV"], 1] = 0
    assert t.check_row(1) != -1


def test_golden_vectors_pin_both_readings_and_the_oracle_proof():
    """tests/golden/mapg2_vectors.json (SELF-golden, tools/gen_golden.py mapg2): twelve messages incl. the edge cases; the C reading
    reproduces the records, the Python reading the cleared points, and the oracle's proof of the records has the committed digest
    (AIR tables, schedule, lookup fill rule, Fiat-Shamir order and FRI pinned against silent drift)"""
    import hashlib
    import json
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "mapg2_vectors.json")))
    recs = np.array(g["records"], dtype=np.uint32)
    assert (_oracle.map_to_g2(recs[:, :16]) == recs).all()
    for r, c, b in zip(g["records"], g["cleared"], g["branch_e1_e2_z"]):
        u = (bn254.u32_to_fq(r[:8]), bn254.u32_to_fq(r[8:16]))
        w = M.witness(u)
        assert [w["e1"], w["e2"], w["z"]] == b
        assert bn254.g2_to_u32(bn254.g2_mul((w["XS"], w["Y"]), bn254.G2_COFACTOR)) == c
    pf = _oracle.stark_prove(3, recs)
    assert (int(len(pf)), int(pf[2]), int(pf[4]), int(pf[5])) == (g["proof"]["words"], g["proof"]["log_n"], g["proof"]["W"], g["proof"]["P"])
    assert hashlib.sha256(pf.tobytes()).hexdigest() == g["proof"]["sha256"]
