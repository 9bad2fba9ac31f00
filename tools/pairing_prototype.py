"""Executable prototype of sipp_amd/csrc/pairing.hip: the same representation (Fq12 = Fq2[w]/(w^6 - xi)), the same affine Miller
loop and the same final exponentiation (easy part, then the hard part by the chain of ark-ec 0.4's Bn::final_exponentiation as
recalled: Fuentes-Castaneda et al.), in big-int Python, checked against oracle/py/bn254.py (plain power by
lambda (p^12 - 1)/r, lambda = 2u(6u^2 + 3u + 1)); the Devegili-Scott-Dahab chain the GPU ran until round 5 is kept as
`final_exp_reduced` (the plain reduced pairing).  Run it to re-verify the formulas: python tools/pairing_prototype.py"""
import sys, random
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.py import bn254 as bn
P, U = bn.P, bn.U
XI = (9, 1)
f2m, f2a, f2s, f2inv, f2conj, f2neg = bn.f2_mul, bn.f2_add, bn.f2_sub, bn.f2_inv, bn.f2_conj, bn.f2_neg
Z2 = (0, 0); O2 = (1, 0)

# ---- T6: Fq2[w]/(w^6 - xi): list of 6 Fq2 ----
def t_one(): return [O2] + [Z2] * 5
def t_mul(a, b):
    d = [Z2] * 11
    for i in range(6):
        for j in range(6):
            d[i + j] = f2a(d[i + j], f2m(a[i], b[j]))
    return [f2a(d[k], f2m(XI, d[k + 6])) if k < 5 else d[k] for k in range(6)]
def t_sqr(a): return t_mul(a, a)
def t_conj(a):  # x -> x^(p^6): w -> -w
    return [a[i] if i % 2 == 0 else f2neg(a[i]) for i in range(6)]
GAMMA = [[bn.f2_pow(XI, (i * (P**k - 1)) // 6) for i in range(6)] for k in range(4)]  # GAMMA[k][i] = xi^(i (p^k - 1)/6)
def t_frob(a, k=1):
    out = []
    for i in range(6):
        c = a[i]
        if k % 2 == 1: c = f2conj(c)
        out.append(f2m(c, GAMMA[k][i]))
    return out
def t_to_c(a):  # -> oracle's 12-coefficient form
    r = [0] * 12
    for i in range(6):
        r[i] = (a[i][0] - 9 * a[i][1]) % P
        r[i + 6] = a[i][1] % P
    return r
def t_from_c(c):
    return [((c[i] + 9 * c[i + 6]) % P, c[i + 6] % P) for i in range(6)]
# Fq6 = Fq2[v]/(v^3 - xi), v = w^2
def s_mul(a, b):
    d = [Z2] * 5
    for i in range(3):
        for j in range(3):
            d[i + j] = f2a(d[i + j], f2m(a[i], b[j]))
    return [f2a(d[0], f2m(XI, d[3])), f2a(d[1], f2m(XI, d[4])), d[2]]
def s_inv(a):
    a0, a1, a2 = a
    c0 = f2s(f2m(a0, a0), f2m(XI, f2m(a1, a2)))
    c1 = f2s(f2m(XI, f2m(a2, a2)), f2m(a0, a1))
    c2 = f2s(f2m(a1, a1), f2m(a0, a2))
    t = f2a(f2m(a0, c0), f2m(XI, f2a(f2m(a2, c1), f2m(a1, c2))))
    ti = f2inv(t)
    return [f2m(c0, ti), f2m(c1, ti), f2m(c2, ti)]
def t_inv(a):
    E = [a[0], a[2], a[4]]; O = [a[1], a[3], a[5]]          # a = E + O w, w^2 = v
    E2 = s_mul(E, E); O2_ = s_mul(O, O)
    vO2 = [f2m(XI, O2_[2]), O2_[0], O2_[1]]                   # v * O^2
    N = [f2s(E2[i], vO2[i]) for i in range(3)]
    Ni = s_inv(N)
    Ei = s_mul(E, Ni); Oi = s_mul(O, Ni)
    return [Ei[0], f2neg(Oi[0]), Ei[1], f2neg(Oi[1]), Ei[2], f2neg(Oi[2])]
def t_pow(a, e):
    r = t_one()
    while e:
        if e & 1: r = t_mul(r, a)
        a = t_sqr(a); e >>= 1
    return r

# ---- Miller loop: the oracle's affine algorithm, in T6 arithmetic ----
ATE = 6 * U + 2
def line(T, Q, Pt):
    if T == Q:
        lam = f2m(bn.f2_scal(f2m(T[0], T[0]), 3), f2inv(bn.f2_scal(T[1], 2)))
    else:
        lam = f2m(f2s(Q[1], T[1]), f2inv(f2s(Q[0], T[0])))
    xp, yp = Pt
    l = [Z2] * 6
    l[0] = (yp % P, 0)
    l[1] = bn.f2_scal(lam, (-xp) % P)
    l[3] = f2s(f2m(lam, T[0]), T[1])
    return l, lam
def step(T, Q, lam):
    x = f2s(f2s(f2m(lam, lam), T[0]), Q[0])
    return (x, f2s(f2m(lam, f2s(T[0], x)), T[1]))
def miller(Pt, Q):
    f = t_one(); T = Q
    for bit in bin(ATE)[3:]:
        l, lam = line(T, T, Pt); f = t_mul(t_sqr(f), l); T = step(T, T, lam)
        if bit == "1":
            l, lam = line(T, Q, Pt); f = t_mul(f, l); T = step(T, Q, lam)
    Q1 = (f2m(f2conj(Q[0]), bn.FROB_X), f2m(f2conj(Q[1]), bn.FROB_Y))
    Q2 = (f2m(f2conj(Q1[0]), bn.FROB_X), f2m(f2conj(Q1[1]), bn.FROB_Y))
    Q2n = (Q2[0], f2neg(Q2[1]))
    l, lam = line(T, Q1, Pt); f = t_mul(f, l); T = step(T, Q1, lam)
    l, lam = line(T, Q2n, Pt); f = t_mul(f, l)
    return f

def exp_x(a):
    return t_pow(a, U)
def exp_neg_x(a):
    return t_conj(exp_x(a))
def final_exp(f):
    """ark-ec 0.4 models/bn/mod.rs final_exponentiation, y_k named as there (recalled)"""
    r = t_mul(t_conj(f), t_inv(f))            # f^(p^6 - 1)
    r = t_mul(t_frob(r, 2), r)                # ^(p^2 + 1)
    y0 = exp_neg_x(r); y1 = t_sqr(y0); y2 = t_sqr(y1); y3 = t_mul(y2, y1)
    y4 = exp_neg_x(y3); y5 = t_sqr(y4); y6 = exp_neg_x(y5)
    y3 = t_conj(y3); y6 = t_conj(y6)
    y7 = t_mul(y6, y4); y8 = t_mul(y7, y3); y9 = t_mul(y8, y1); y10 = t_mul(y8, y4); y11 = t_mul(y10, r)
    y12 = t_frob(y9, 1); y13 = t_mul(y12, y11)
    y8 = t_frob(y8, 2); y14 = t_mul(y8, y13)
    y15 = t_frob(t_mul(t_conj(r), y9), 3)
    return t_mul(y15, y14)
def final_exp_reduced(f):
    # easy part: f^((p^6 - 1)(p^2 + 1))
    g = t_mul(t_conj(f), t_inv(f))
    g = t_mul(t_frob(g, 2), g)
    # hard part (Devegili-Scott-Dahab): g^((p^4 - p^2 + 1)/r)
    fx = exp_x(g); fx2 = exp_x(fx); fx3 = exp_x(fx2)
    y0 = t_mul(t_mul(t_frob(g, 1), t_frob(g, 2)), t_frob(g, 3))
    y1 = t_conj(g)
    y2 = t_frob(fx2, 2)
    y3 = t_conj(t_frob(fx, 1))
    y4 = t_conj(t_mul(fx, t_frob(fx2, 1)))
    y5 = t_conj(fx2)
    y6 = t_conj(t_mul(fx3, t_frob(fx3, 1)))
    t0 = t_mul(t_mul(t_sqr(y6), y4), y5)
    t1 = t_mul(t_mul(y3, y5), t0)
    t0 = t_mul(t0, y2)
    t1 = t_sqr(t_mul(t_sqr(t1), t0))
    t0 = t_mul(t1, y1)
    t1 = t_mul(t1, y0)
    t0 = t_sqr(t0)
    return t_mul(t0, t1)

if __name__ == "__main__":
    rnd = random.Random(5)
    a = [(rnd.randrange(P), rnd.randrange(P)) for _ in range(6)]
    b = [(rnd.randrange(P), rnd.randrange(P)) for _ in range(6)]
    assert t_to_c(t_mul(a, b)) == bn.f12_mul(t_to_c(a), t_to_c(b)), "mul / basis"
    assert t_from_c(t_to_c(a)) == a
    assert t_mul(a, t_inv(a)) == t_one(), "inv"
    for k in (1, 2, 3):
        assert t_to_c(t_frob(a, k)) == bn.f12_pow(t_to_c(a), P**k), "frob %d" % k
    assert t_to_c(t_conj(a)) == bn.f12_pow(t_to_c(a), P**6), "conj"
    s, t = rnd.randrange(1, bn.R), rnd.randrange(1, bn.R)
    Pt, Q = bn.g1_mul(bn.G1, s), bn.g2_mul(bn.G2, t)
    m = miller(Pt, Q)
    assert t_to_c(m) == bn.miller_loop(Pt, Q), "miller"
    e = final_exp(m)
    assert t_to_c(e) == bn.pairing(Pt, Q), "final exp"
    assert t_to_c(final_exp_reduced(m)) == bn.pairing_reduced(Pt, Q), "reduced final exp"
    assert t_to_c(t_pow(final_exp_reduced(m), bn.ARK_MULTIPLIER)) == bn.pairing(Pt, Q), "ark value = reduced ^ lambda"
    print("prototype ok")


# ---- projective Miller loop (no inversion per step): what miller_coop_kernel in pairing.hip runs ----
# T = (X, Y, Z) homogeneous, x = X/Z, y = Y/Z.  Lines are scaled by Fq2 factors (2 Y Z^2 for a tangent, x_Q Z - X for a chord),
# which the final exponentiation removes; the Miller VALUE therefore differs from the affine one, the pairing does not.
def f2sq(a): return f2m(a, a)
def f2sc(a, s): return bn.f2_scal(a, s % P)
def proj_double(T, Pt):
    X, Y, Z = T
    N = f2sc(f2sq(X), 3); D = f2sc(f2m(Y, Z), 2)
    D2 = f2sq(D); D3 = f2m(D2, D); XD2 = f2m(X, D2)
    W = f2s(f2m(f2sq(N), Z), f2sc(XD2, 2))
    l0 = f2sc(f2m(D, Z), Pt[1]); l1 = f2sc(f2m(N, Z), -Pt[0]); l3 = f2s(f2m(N, X), f2m(D, Y))
    T3 = (f2m(D, W), f2s(f2m(N, f2s(XD2, W)), f2m(Y, D3)), f2m(D3, Z))
    return T3, (l0, l1, l3)
def proj_add(T, Q, Pt):
    X, Y, Z = T; xq, yq = Q
    N = f2s(f2m(yq, Z), Y); D = f2s(f2m(xq, Z), X)
    D2 = f2sq(D); D3 = f2m(D2, D); E = f2m(D2, Z); xqE = f2m(xq, E)
    W = f2s(f2s(f2m(f2sq(N), Z), f2m(X, D2)), xqE)
    Z3 = f2m(D3, Z)
    l0 = f2sc(D, Pt[1]); l1 = f2sc(N, -Pt[0]); l3 = f2s(f2m(N, xq), f2m(D, yq))
    T3 = (f2m(D, W), f2s(f2m(N, f2s(xqE, W)), f2m(yq, Z3)), Z3)
    return T3, (l0, l1, l3)
def mul_line(f, l):
    L = [l[0], l[1], Z2, l[2], Z2, Z2]
    return t_mul(f, L)
def miller_proj(Pt, Q):
    f = t_one(); T = (Q[0], Q[1], O2)
    for bit in bin(ATE)[3:]:
        T, l = proj_double(T, Pt); f = mul_line(t_sqr(f), l)
        if bit == "1":
            T, l = proj_add(T, Q, Pt); f = mul_line(f, l)
    Q1 = (f2m(f2conj(Q[0]), bn.FROB_X), f2m(f2conj(Q[1]), bn.FROB_Y))
    Q2 = (f2m(f2conj(Q1[0]), bn.FROB_X), f2m(f2conj(Q1[1]), bn.FROB_Y))
    T, l = proj_add(T, Q1, Pt); f = mul_line(f, l)
    T, l = proj_add(T, (Q2[0], f2neg(Q2[1])), Pt); f = mul_line(f, l)
    return f


if __name__ == "__main__":
    rnd = random.Random(9)
    for _ in range(2):
        s, t = rnd.randrange(1, bn.R), rnd.randrange(1, bn.R)
        Pt, Q = bn.g1_mul(bn.G1, s), bn.g2_mul(bn.G2, t)
        assert t_to_c(final_exp(miller_proj(Pt, Q))) == bn.pairing(Pt, Q), "projective Miller loop"
    print("projective prototype ok")
