"""The AIRs (tools/air_gen.py specification) on the CPU oracle: traces built from native SIPP obligations
satisfy every program constraint on every row; tampering is detected; wrong claimed outputs are refused."""
import numpy as np
import pytest

from oracle.py import sipp_native as sn
from tests import _oracle


@pytest.fixture(scope="module")
def ios4():
    A, B = sn.synthetic_inputs(4, 7)
    proof = sn.sipp_prove_native(A, B)
    ok, st, obl = sn.sipp_verify_native(A, B, proof, check_final_pairing=False)
    return sn.io_records(obl)


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_trace_satisfies_constraints(ios4, kind):
    ios = ios4[kind]
    t = _oracle.Trace(kind, ios)
    n = 1 << t.log_n
    assert t.air.table_bits == 8 and n == 512 * t.num_io
    rows = list(range(0, 70)) + list(range(500, 530)) + list(range(n - 3, n)) + [1023, 1024, 1087, 1088]
    for r in rows:
        if r < n:
            assert t.check_row(r) == -1, (kind, r)
    arr = t.array()
    # range table and lookups
    tb = 1 << t.air.table_bits
    assert (arr[0] == np.minimum(np.arange(n), tb - 1)).all()
    nm, nc, cb = t.air.n_main, t.air.n_checked, t.air.checked_base
    for j in (0, nc // 2, nc - 1):
        col, pin, ptab = arr[cb + j], arr[nm + j], arr[nm + nc + j]
        assert col.max() < tb
        assert (np.sort(col) == pin).all()
        assert (np.sort(ptab) == np.sort(arr[0])).all()
        first = np.concatenate([[True], pin[1:] != pin[:-1]])
        assert (ptab[first] == pin[first]).all()


def test_tamper_detected(ios4):
    t = _oracle.Trace(0, ios4[0])
    arr = t.array()
    r = 37
    assert t.check_row(r) == -1
    col = t.air.checked_base + 5          # a limb of lambda
    arr[col, r] ^= 1
    assert t.check_row(r) != -1
    arr[col, r] ^= 1
    arr[1, r] += 1                         # accumulator limb: breaks the transition from row r-1 / gadgets of r
    assert t.check_row(r) != -1 or t.check_row(r - 1) != -1


def test_wrong_output_refused(ios4):
    ios = ios4[0].copy()
    ios[1, 55] ^= 1                        # last word of the claimed output of IO 1
    with pytest.raises(RuntimeError):
        _oracle.Trace(0, ios)


def crafted_g1_records():
    """edge-case obligations: exp = 0 (out = offset), exp = 1, exp = r - 1, exp = 2^255-ish bit patterns"""
    from oracle.py import bn254 as bn
    from oracle.py import sipp_native as sn
    x = bn.g1_mul(bn.G1, 0x1234567)
    off = bn.g1_mul(bn.G1, 0x89abcdef)
    recs = []
    for e in (0, 1, bn.R - 1, (1 << 253) + 5):
        out = bn.g1_add(off, bn.g1_mul(x, e)) if e else off
        recs.append(bn.g1_to_u32(x) + bn.g1_to_u32(off) + sn.exp_to_u32(e) + bn.g1_to_u32(out))
    return np.array(recs, dtype=np.uint32)


def crafted_g2_records():
    """the same edge exponents for G2ExpStark"""
    from oracle.py import bn254 as bn
    from oracle.py import sipp_native as sn
    x = bn.g2_mul(bn.G2, 0x7654321)
    off = bn.g2_mul(bn.G2, 0xfedcba98)
    recs = []
    for e in (0, 1, bn.R - 1, (1 << 253) + 5):
        out = bn.g2_add(off, bn.g2_mul(x, e)) if e else off
        recs.append(bn.g2_to_u32(x) + bn.g2_to_u32(off) + sn.exp_to_u32(e) + bn.g2_to_u32(out))
    return np.array(recs, dtype=np.uint32)


def crafted_fq12_records():
    """edge exponents for Fq12ExpStark: out = offset * x^exp in the pairing target group"""
    from oracle.py import bn254 as bn
    from oracle.py import sipp_native as sn
    g = bn.pairing(bn.G1, bn.G2)
    x = bn.f12_pow(g, 0x1357)
    off = bn.f12_pow(g, 0x2468)
    recs = []
    for e in (0, 1, bn.R - 1, (1 << 253) + 5):
        out = bn.f12_mul(off, bn.f12_pow(x, e)) if e else off
        recs.append(bn.f12_to_u32(x) + bn.f12_to_u32(off) + sn.exp_to_u32(e) + bn.f12_to_u32(out))
    return np.array(recs, dtype=np.uint32)


def test_edge_exponents_g2_fq12_traces_satisfy_the_air():
    for kind, recs in ((1, crafted_g2_records()), (2, crafted_fq12_records())):
        t = _oracle.Trace(kind, recs)
        for r in list(range(0, 4)) + [510, 511, 512, 1023, 1024, 1535, 2047]:
            assert t.check_row(r) == -1, (kind, r)


def test_edge_exponents_and_degenerate_inputs():
    recs = crafted_g1_records()
    t = _oracle.Trace(0, recs)
    for r in list(range(0, 4)) + [510, 511, 512, 1023, 1024, 1535, 2047]:
        assert t.check_row(r) == -1, r
    # x == offset with an odd exponent: the very first addition is R + P with R == P -> not provable, loudly
    from oracle.py import bn254 as bn
    from oracle.py import sipp_native as sn
    x = bn.g1_mul(bn.G1, 77)
    bad = np.array([bn.g1_to_u32(x) + bn.g1_to_u32(x) + sn.exp_to_u32(3) + bn.g1_to_u32(bn.g1_mul(x, 4))], dtype=np.uint32)
    with pytest.raises(RuntimeError):
        _oracle.Trace(0, bad)
