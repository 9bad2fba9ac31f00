#!/usr/bin/env python3
"""Row schedule of the final-pairing AIR (API kind 6, tools/air_gen.py::build_pairing): ONE optimal ate pairing e(P, Q) in 512 trace
rows -- what the reference's BLS example asks of its outer circuit at src/bin/bls_aggregation.rs:76-77
(`pairing_circuit(final_A, final_B)` connected to `final_Z`; plonky2-bn254-pairing @ fe5c3a8, not vendored), here as a STARK
obligation like the three exponentiations.  The value is arkworks' (oracle/py/bn254.py: final exponent lambda (p^12 - 1)/r), by
ark-ec 0.4's own chain: easy part, then y0 .. y16 of Bn::final_exponentiation (recalled; tools/pairing_prototype.py::final_exp).

A row has two units:
  * the Fq12 unit: C = A * B  (MUL; B may be the line of the same row's G2 unit),  A * C = 1  (INV: C is the inverse),
    C_i = conj^c(A_i) * G_i  (FROB: coefficient-wise product with the row's constants G: a Frobenius map or a conjugation),
    A, B = one of NREG registers; "load" = the register that takes C over at the end of the row;
  * the G2 unit: a tangent step at T (TG), a chord step through T and Q / pi(Q) / -pi^2(Q) (CH0 / CH1 / CH2), both with the line
    coefficients at P, or the Frobenius images of Q (FQ, row 0).
The schedule is DATA: emitted into data/air_tables.h for the two trace generators (oracle/pairing.c, sipp_amd/csrc/pairing_stark.hip)
and turned into the periodic columns of the AIR.  `simulate` runs it in big-int Python (the executable specification;
tests/test_oracle_pairing_air.py compares its result with oracle/py/bn254.py::pairing)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.py import bn254 as bn  # noqa: E402

P, U = bn.P, bn.U
ROWS = 512
F_IDLE, F_MUL, F_LINE, F_INV, F_FROB = 0, 1, 2, 3, 4
G_IDLE, G_TG, G_CH0, G_CH1, G_CH2, G_FQ = 0, 1, 2, 3, 4, 5
# constant vectors G (six Fq2 each): ONE (row 0 loads it into the Miller accumulator; the INV row's right-hand side), CONJ
# ((-1)^i: x -> x^(p^6)), and the Frobenius constants gamma_k,i = xi^(i (p^k - 1)/6), k = 1, 2, 3
GC_ONE, GC_CONJ, GC_F1, GC_F2, GC_F3 = 0, 1, 2, 3, 4
XI = (9, 1)
GAMMA = [[bn.f2_pow(XI, (i * (P**k - 1)) // 6) for i in range(6)] for k in range(4)]
G_CONSTS = [
    [(1, 0)] + [(0, 0)] * 5,
    [((P - 1) if i & 1 else 1, 0) for i in range(6)],
    GAMMA[1], GAMMA[2], GAMMA[3],
]
G_CONJ_COEF = [0, 0, 1, 0, 1]          # 1: the coefficients are conjugated before the product (odd Frobenius powers)


def virtual_program():
    """[(fop, a, b, gconst, dst, gop)] over virtual values (names); a / b / dst None where unused"""
    rows = []
    cnt = [0]

    def new(prefix):
        cnt[0] += 1
        return "%s%d" % (prefix, cnt[0])

    def emit(fop, a=None, b=None, gc=-1, dst=None, gop=G_IDLE):
        rows.append(dict(fop=fop, a=a, b=b, gc=gc, dst=dst, gop=gop))
        return dst

    # ---- Miller loop (oracle/py/bn254.py::miller_loop: affine steps, plain binary expansion of 6u + 2) ----
    emit(F_IDLE, gc=GC_ONE, dst="f0", gop=G_FQ)          # row 0: f <- 1 (from the constants), T <- Q, Q1 / Q2N from the G2 unit
    f = "f0"
    first = True
    for bit in bin(bn.ATE)[3:]:
        if not first:
            f = emit(F_MUL, f, f, dst=new("f"))
        first = False
        f = emit(F_LINE, f, dst=new("f"), gop=G_TG)
        if bit == "1":
            f = emit(F_LINE, f, dst=new("f"), gop=G_CH0)
    f = emit(F_LINE, f, dst=new("f"), gop=G_CH1)
    f = emit(F_LINE, f, dst=new("f"), gop=G_CH2)
    # ---- easy part ----
    g = emit(F_INV, f, gc=GC_ONE, dst=new("g"))
    fc = emit(F_FROB, f, gc=GC_CONJ, dst=new("fc"))
    r0 = emit(F_MUL, fc, g, dst=new("r"))
    x = emit(F_FROB, r0, gc=GC_F2, dst=new("x"))
    r = emit(F_MUL, x, r0, dst=new("r"))

    def exp_u(base):
        acc = None
        for bit in bin(U)[3:]:
            acc = emit(F_MUL, acc or base, acc or base, dst=new("e"))
            if bit == "1":
                acc = emit(F_MUL, acc, base, dst=new("e"))
        return acc

    def conj(a):
        return emit(F_FROB, a, gc=GC_CONJ, dst=new("c"))

    def mul(a, b):
        return emit(F_MUL, a, b, dst=new("y"))

    # ---- hard part: ark-ec 0.4 Bn::final_exponentiation, y_k as named there ----
    y0 = conj(exp_u(r))
    y1 = mul(y0, y0)
    y2 = mul(y1, y1)
    y3 = mul(y2, y1)
    y4 = conj(exp_u(y3))
    y5 = mul(y4, y4)
    y6c = exp_u(y5)                 # = conj(y6)
    y3c = conj(y3)
    y7 = mul(y6c, y4)
    y8 = mul(y7, y3c)
    y9 = mul(y8, y1)
    y10 = mul(y8, y4)
    y11 = mul(y10, r)
    y12 = emit(F_FROB, y9, gc=GC_F1, dst=new("y"))
    y13 = mul(y12, y11)
    y8f = emit(F_FROB, y8, gc=GC_F2, dst=new("y"))
    y14 = mul(y8f, y13)
    rc = conj(r)
    y15 = mul(rc, y9)
    y15f = emit(F_FROB, y15, gc=GC_F3, dst=new("y"))
    mul(y15f, y14)
    return rows


def allocate(rows):
    """virtual values -> registers.  A register may be reloaded at the end of the row that reads its value for the last time."""
    last = {}
    for t, r in enumerate(rows):
        for v in (r["a"], r["b"]):
            if v is not None:
                last[v] = t
    result = rows[-1]["dst"]
    last[result] = ROWS                      # the result stays until the block's last row
    reg_of, busy_until, nreg = {}, [], 0
    out = []
    for t, r in enumerate(rows):
        dst = r["dst"]
        ra = reg_of[r["a"]] if r["a"] is not None else -1
        rb = reg_of[r["b"]] if r["b"] is not None else -1
        rd = -1
        if dst is not None:
            free = [k for k in range(nreg) if busy_until[k] <= t]
            if free:
                rd = free[0]
            else:
                rd = nreg
                nreg += 1
                busy_until.append(0)
            busy_until[rd] = last.get(dst, t)          # a value nobody reads frees its register at once
            reg_of[dst] = rd
        out.append(dict(fop=r["fop"], ra=ra, rb=rb, gc=r["gc"], rd=rd, gop=r["gop"]))
    while len(out) < ROWS:
        out.append(dict(fop=F_IDLE, ra=-1, rb=-1, gc=-1, rd=-1, gop=G_IDLE))
    assert len(out) == ROWS
    return out, nreg, reg_of[result]


SCHEDULE, NREG, RESULT_REG = allocate(virtual_program())
N_ACTIVE = len(virtual_program())


# ---------------------------------------------------------------------------------------------------------------------------
# executable specification: the schedule in big-int Python (tower basis Fq2[w]/(w^6 - xi), six Fq2 coefficients)
f2m, f2a, f2s, f2neg, f2conj, f2inv = bn.f2_mul, bn.f2_add, bn.f2_sub, bn.f2_neg, bn.f2_conj, bn.f2_inv
Z2 = (0, 0)


def t_mul(a, b):
    d = [Z2] * 11
    for i in range(6):
        for j in range(6):
            d[i + j] = f2a(d[i + j], f2m(a[i], b[j]))
    return [f2a(d[k], f2m(XI, d[k + 6])) if k < 5 else d[k] for k in range(6)]


def t_inv(a):
    c = bn.f12_inv(t_to_c(a))
    return t_from_c(c)


def t_to_c(a):
    r = [0] * 12
    for i in range(6):
        r[i] = (a[i][0] - 9 * a[i][1]) % P
        r[i + 6] = a[i][1] % P
    return r


def t_from_c(c):
    return [((c[i] + 9 * c[i + 6]) % P, c[i + 6] % P) for i in range(6)]


def simulate(Pt, Q, trace=None):
    """runs SCHEDULE on (P, Q); returns the result as 12 MyFq12 coefficients.  trace (a list) receives one dict per row with
    every primary value of that row (what the trace generators write): regs, A, B, C, G, T, QS, S0..S4, Q1, Q2N."""
    regs = [[Z2] * 6 for _ in range(NREG)]
    T = (Z2, Z2)
    Q1 = (Z2, Z2)
    Q2N = (Z2, Z2)
    xp, yp = Pt
    zero6 = [Z2] * 6
    for t, r in enumerate(SCHEDULE):
        S = [Z2] * 5
        QS = (Z2, Z2)
        gop = r["gop"]
        if gop == G_FQ:
            S[0] = f2m(f2conj(Q[0]), bn.FROB_X)
            S[1] = f2m(f2conj(Q[1]), bn.FROB_Y)
            S[2] = f2m(f2conj(S[0]), bn.FROB_X)
            S[3] = f2neg(f2m(f2conj(S[1]), bn.FROB_Y))
        elif gop != G_IDLE:
            if gop == G_TG:
                den = bn.f2_scal(T[1], 2)
                num = bn.f2_scal(f2m(T[0], T[0]), 3)
                xb = T[0]
            else:
                QS = Q if gop == G_CH0 else Q1 if gop == G_CH1 else Q2N
                den = f2s(QS[0], T[0])
                num = f2s(QS[1], T[1])
                xb = QS[0]
            if den == Z2:
                raise ValueError("degenerate step at row %d" % t)
            lam = f2m(num, f2inv(den))
            S[0] = lam
            S[1] = f2s(f2s(f2m(lam, lam), T[0]), xb)
            S[2] = f2s(f2m(lam, f2s(T[0], S[1])), T[1])
            S[3] = f2neg(bn.f2_scal(lam, xp))                 # L1N = -lam x_P
            S[4] = f2s(f2m(lam, T[0]), T[1])                  # L3 = lam x_T - y_T
        fop = r["fop"]
        G = G_CONSTS[r["gc"]] if r["gc"] >= 0 else zero6
        A = regs[r["ra"]] if r["ra"] >= 0 else zero6
        if fop == F_MUL:
            B = regs[r["rb"]]
            C = t_mul(A, B)
        elif fop == F_LINE:
            B = [(yp % P, 0), S[3], Z2, S[4], Z2, Z2]
            C = t_mul(A, B)
        elif fop == F_INV:
            C = t_inv(A)
            B = C
        elif fop == F_FROB:
            B = zero6
            C = [f2m(f2conj(A[i]) if G_CONJ_COEF[r["gc"]] else A[i], G[i]) for i in range(6)]
        else:
            B = zero6
            C = zero6
        if trace is not None:
            trace.append(dict(regs=[list(x) for x in regs], A=A, B=B, C=C, G=G, T=T, QS=QS, S=S, Q1=Q1, Q2N=Q2N))
        # end of row: loads
        if r["rd"] >= 0:
            regs[r["rd"]] = list(G) if fop == F_IDLE else list(C)
        if gop == G_FQ:
            T, Q1, Q2N = Q, (S[0], S[1]), (S[2], S[3])
        elif gop in (G_TG, G_CH0, G_CH1):
            T = (S[1], S[2])
    return t_to_c(regs[RESULT_REG])


if __name__ == "__main__":
    import random
    from collections import Counter
    print("active rows", N_ACTIVE, "registers", NREG, "result in register", RESULT_REG)
    print("fq12 ops", Counter(r["fop"] for r in SCHEDULE), "g2 ops", Counter(r["gop"] for r in SCHEDULE))
    rnd = random.Random(11)
    for _ in range(2):
        s, t = rnd.randrange(1, bn.R), rnd.randrange(1, bn.R)
        Pt, Q = bn.g1_mul(bn.G1, s), bn.g2_mul(bn.G2, t)
        assert simulate(Pt, Q) == bn.pairing(Pt, Q)
    assert simulate(bn.G1, bn.G2) == bn.pairing(bn.G1, bn.G2)
    print("schedule ok: simulate == oracle/py/bn254.py::pairing")
