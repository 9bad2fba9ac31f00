"""Pins the BN254 restatement (Python big-int) by the identities the domain offers, and the C Fq/Fq12
arithmetic against it.  CPU only."""
import ctypes as C

import numpy as np

from oracle.py import bn254 as bn


def test_generators_on_curve_and_order():
    assert bn.g1_on_curve(bn.G1) and bn.g2_on_curve(bn.G2)
    assert bn.g1_mul(bn.G1, bn.R) is None
    assert bn.g2_mul(bn.G2, bn.R) is None


def test_fq12_form_matches_tower_identity():
    # u = w^6 - 9 must satisfy u^2 = -1
    u = [(-9) % bn.P, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0]
    assert bn.f12_mul(u, u) == [bn.P - 1] + [0] * 11
    a = [(i * 7919 + 13) % bn.P for i in range(12)]
    assert bn.f12_mul(a, bn.f12_inv(a)) == bn.F12_ONE


def test_pairing_bilinear_nondegenerate():
    e = bn.pairing(bn.G1, bn.G2)
    assert e != bn.F12_ONE
    assert bn.f12_pow(e, bn.R) == bn.F12_ONE
    a, b = 0x1234567, 0x89abcdef1
    lhs = bn.pairing(bn.g1_mul(bn.G1, a), bn.g2_mul(bn.G2, b))
    assert lhs == bn.f12_pow(e, a * b % bn.R)
    assert bn.multi_pairing([bn.G1, bn.g1_neg(bn.G1)], [bn.G2, bn.G2]) == bn.F12_ONE


def test_c_fq_and_fq12_match_python(oracle):
    oracle.fq_init()

    class Fq(C.Structure):
        _fields_ = [("l", C.c_uint64 * 4)]

    class Fq12(C.Structure):
        _fields_ = [("c", Fq * 12)]
    oracle.fq_mul.restype = Fq
    oracle.fq_mul.argtypes = [Fq, Fq]
    oracle.fq_inv.restype = Fq
    oracle.fq_inv.argtypes = [Fq]
    oracle.fq12_mul.restype = Fq12
    oracle.fq12_mul.argtypes = [C.POINTER(Fq12), C.POINTER(Fq12)]

    def mk(v):
        f = Fq()
        for i in range(4):
            f.l[i] = (v >> (64 * i)) & (2**64 - 1)
        return f

    def val(f):
        return sum(int(f.l[i]) << (64 * i) for i in range(4))
    rng = np.random.default_rng(3)
    for _ in range(20):
        a = int.from_bytes(rng.bytes(32), "little") % bn.P
        b = int.from_bytes(rng.bytes(32), "little") % bn.P
        assert val(oracle.fq_mul(mk(a), mk(b))) == a * b % bn.P
        assert val(oracle.fq_inv(mk(a))) == bn.inv(a)
    assert val(oracle.fq_mul(mk(bn.P - 1), mk(bn.P - 1))) == 1
    x = [int.from_bytes(rng.bytes(32), "little") % bn.P for _ in range(12)]
    y = [int.from_bytes(rng.bytes(32), "little") % bn.P for _ in range(12)]
    fx, fy = Fq12(), Fq12()
    for k in range(12):
        fx.c[k] = mk(x[k])
        fy.c[k] = mk(y[k])
    r = oracle.fq12_mul(C.byref(fx), C.byref(fy))
    assert [val(r.c[k]) for k in range(12)] == bn.f12_mul(x, y)


def test_fq2_sqrt_and_cofactor_clearing():
    """E'(Fp2) has r (2p - r) points: a twist point found by try-and-increment is (almost surely) outside the r-torsion, its
    cofactor multiple is inside (what bls_aggregation.rs:103-106 does with mapped messages)"""
    from oracle.py import bn254 as bn
    for a in [(4, 0), (5, 7), (123456789, 987654321), (0, 1)]:
        sq = bn.f2_mul(a, a)
        r = bn.f2_sqrt(sq)
        assert r is not None and bn.f2_mul(r, r) == sq
    assert sum(bn.f2_sqrt((k, 1)) is None for k in range(40)) > 5          # about half of the elements are non-squares
    for seed in (1, 2, 3):
        pt = bn.g2_twist_point(seed)
        assert bn.g2_on_curve(pt)
        assert bn.g2_mul(pt, bn.R) is not None                               # not in the r-torsion
        assert bn.g2_mul(pt, bn.R * bn.G2_COFACTOR) is None                  # the group order kills it
        cleared = bn.g2_clear_cofactor(pt)
        assert cleared is not None and bn.g2_on_curve(cleared) and bn.g2_mul(cleared, bn.R) is None
