"""Map an Fp2 element to a point of the BN254 twist E'(Fp2): y^2 = x^3 + 3/(9+u)  -- second (Python) reading.

TEST INFRASTRUCTURE ONLY; PARITY UNPINNED.  The reference reaches the map through
`plonky2_bn254::curves::map_to_g2::map_to_g2_without_cofactor_mul` (src/bin/bls_aggregation.rs:21, :102) and its STARK
through `starky_bn254::curves::g2::batch_map_to_g2::batch_map_to_g2_circuit` (:31, :65); neither crate is under
/root/reference.  What is restated here is the PUBLISHED algorithm those crates are recalled to follow: the
Shallue - van de Woestijne map of RFC 9380 section 6.6.1 / appendix F.1 with Z = 1 (the smallest Z that `find_z_svdw`
accepts for this curve), `sgn0` of section 4.1 (m = 2) and `is_square(0) = true`.  oracle/mapg2.c is the C reading;
tests/test_oracle_mapg2.py compares the two value for value.
"""
from bn254 import P, B2, XI, G2_COFACTOR, f2_add, f2_sub, f2_mul, f2_neg, f2_inv, f2_scal, f2_sqrt, g2_mul, g2_on_curve, inv

Z = (1, 0)
ONE = (1, 0)


def g(x):
    return f2_add(f2_mul(f2_mul(x, x), x), B2)


def sgn0(a):
    """RFC 9380 section 4.1, m = 2: the parity of the first non-zero coordinate"""
    s0, z0 = a[0] & 1, a[0] == 0
    return s0 | (z0 & (a[1] & 1))


def is_square(a):
    return f2_sqrt(a) is not None


def sqrt_even(a):
    """THE square root with sgn0 = 0 (None for a non-square): the deterministic choice both provers make for witness roots"""
    r = f2_sqrt(a)
    if r is None:
        return None
    return f2_neg(r) if sgn0(r) else r


C1 = g(Z)                                                        # g(Z)
C2 = f2_scal(f2_neg(Z), inv(2))                                  # -Z / 2
C3 = sqrt_even(f2_neg(f2_mul(C1, f2_scal(f2_mul(Z, Z), 3))))     # sqrt(-g(Z) (3 Z^2 + 4 A)), sgn0 = 0   (A = 0)
C4 = f2_mul(f2_scal(f2_neg(C1), 4), f2_inv(f2_scal(f2_mul(Z, Z), 3)))   # -4 g(Z) / (3 Z^2 + 4 A)
assert C3 is not None


def witness(u):
    """every intermediate value of the map, named as the columns of the AIR (tools/air_gen.py::build_map_g2); "z" = 1 where the
    map's one inversion meets zero (u^2 g(Z) = +-1, four values of u): inv0(0) = 0"""
    w = {}
    w["T1"] = f2_mul(u, u)
    w["TV1"] = f2_mul(C1, w["T1"])
    tv2 = f2_add(ONE, w["TV1"])
    tv1 = f2_sub(ONE, w["TV1"])
    w["W"] = f2_mul(tv1, tv2)
    w["z"] = int(w["W"] == (0, 0))
    w["TV3"] = (0, 0) if w["z"] else f2_inv(w["W"])
    w["A4"] = f2_mul(u, tv1)
    w["B4"] = f2_mul(w["A4"], w["TV3"])
    tv4 = f2_mul(w["B4"], C3)
    w["X1"] = f2_sub(C2, tv4)
    w["X2"] = f2_add(C2, tv4)
    w["S1"] = f2_mul(w["X1"], w["X1"])
    w["GX1"] = f2_add(f2_mul(w["S1"], w["X1"]), B2)
    w["S2"] = f2_mul(w["X2"], w["X2"])
    w["GX2"] = f2_add(f2_mul(w["S2"], w["X2"]), B2)
    w["D"] = f2_mul(tv2, tv2)
    w["E"] = f2_mul(w["D"], w["TV3"])
    w["F"] = f2_mul(w["E"], w["E"])
    w["X3"] = f2_add(f2_mul(w["F"], C4), Z)
    w["S3"] = f2_mul(w["X3"], w["X3"])
    w["GX3"] = f2_add(f2_mul(w["S3"], w["X3"]), B2)
    e1 = is_square(w["GX1"])
    e2 = (not e1) and is_square(w["GX2"])
    w["e1"], w["e2"] = int(e1), int(e2)
    # a non-square times the non-residue 9 + u is a square: the witness of "not a square"
    w["M1"] = (0, 0) if e1 else w["GX1"]
    w["M2"] = (0, 0) if (e1 or e2) else w["GX2"]
    w["N1"] = sqrt_even(f2_mul(XI, w["M1"]))
    w["N2"] = sqrt_even(f2_mul(XI, w["M2"]))
    w["XS"] = w["X1"] if e1 else w["X2"] if e2 else w["X3"]
    w["GXS"] = w["GX1"] if e1 else w["GX2"] if e2 else w["GX3"]
    y = sqrt_even(w["GXS"])
    assert y is not None and w["N1"] is not None and w["N2"] is not None
    if sgn0(u) != sgn0(y):
        y = f2_neg(y)
    w["Y"] = y
    return w


def map_to_g2_without_cofactor_mul(u):
    w = witness(u)
    return (w["XS"], w["Y"])


def map_to_g2(u):
    """what src/bin/bls_aggregation.rs:102 computes per message: the map, then the cofactor"""
    return g2_mul(map_to_g2_without_cofactor_mul(u), G2_COFACTOR)


if __name__ == "__main__":
    import random
    rnd = random.Random(1)
    for _ in range(8):
        u = (rnd.randrange(P), rnd.randrange(P))
        q = map_to_g2_without_cofactor_mul(u)
        assert g2_on_curve(q) and sgn0(q[1]) == sgn0(u)
    print("C1", C1, "\nC2", C2, "\nC3", C3, "\nC4", C4)
