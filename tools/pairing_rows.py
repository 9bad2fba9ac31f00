#!/usr/bin/env python3
"""Row program of the NARROW final-pairing AIR (API kind 6 since round 6b; tools/air_gen.py::build_pairing): the operation schedule of
tools/pairing_sched.py (458 operations: Miller loop, easy part, ark-ec's hard-part chain) expanded into ONE MODULAR IDENTITY PER ROW --
2^13 rows per pairing, ~2.7 k trace columns -- instead of 22 identities per row on 512 rows and 10.5 k columns: the proof is five times
smaller (what a recursive verifier has to open) and its leaves are hashed in a third of the time.

A row evaluates the one gadget of the AIR on one Fq component:
  FMUL t    RES = component t of A * B            (A, B: operand cells multiplexed from the six Fq12 registers; B may be the current line
                                                    y_P - lam x_P w + (lam x_T - y_T) w^3 or, on FINVC rows, the accumulator CACC)
  FFROB t   RES = A_(k,0) u + A_(k,1) v            (k = t // 2; (u, v) = the row's constants: a Frobenius coefficient, signs folded in)
  FINVW t   RES = component t of 1 / A             (a free, range-checked witness, collected in CACC)
  FINVC t   component t of A * CACC = 1            (no result)
  FCOPY t   RES = component t of the line          (the Miller accumulator starts as the first line: 1 * l)
  FCOMMIT   no identity; the register named by `ld` takes CACC over at the end of the row
  GW c      RES = component c of the slope (free witness); GSL c: component c of the slope identity (tangent: 2 lam y_T = 3 x_T^2,
            chord: lam (x_QS - x_T) = y_QS - y_T); GX3 / GY3 / GL1 / GL3 c: component c of x3, y3, -lam x_P, lam x_T - y_T
  GFQ t     RES = component t % 2 of pi(Q).x, pi(Q).y, pi^2(Q).x, -pi^2(Q).y (t // 2 = 0 .. 3)
Every RES lands in a register cell at the end of its row (CACC[t] for the Fq12 unit, S[slot][c] for the G2 unit); T takes (S1, S2) at the
end of a step, Q at the end of the GFQ rows.  `simulate_rows` is the executable specification (big-int Python) the trace generators
(oracle/pairing.c, sipp_amd/csrc/pairing.hip) are compared with."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pairing_sched as PS  # noqa: E402

bn = PS.bn
P = PS.P
LOG_ROWS = 13
ROWS = 1 << LOG_ROWS
(T_IDLE, T_FMUL, T_FFROB, T_FINVW, T_FINVC, T_FCOPY, T_FCOMMIT, T_GW, T_GSL, T_GX3, T_GY3, T_GL1, T_GL3, T_GFQ) = range(14)
TYPE_NAMES = "IDLE FMUL FFROB FINVW FINVC FCOPY FCOMMIT GW GSL GX3 GY3 GL1 GL3 GFQ".split()
B_REG, B_LINE, B_CACC = 0, 1, 2
SK_NONE, SK_TANGENT, SK_CHORD = 0, 1, 2
END_STEP, END_FQ = 1, 2
# fields of a row descriptor (int8 each): AIR_PAIRING_ROWPROG[row][...]
F_TYP, F_T, F_RA, F_RB, F_BSEL, F_GC, F_LD, F_SK, F_CHM, F_END, N_FIELDS = range(11)


def _frob_pairs():
    """row constants (u, v) of the FFROB rows: RES = A_(k,0) u + A_(k,1) v for constant vector g, coefficient k, component c"""
    pairs, index = [(0, 0)], {}
    for g in range(len(PS.G_CONSTS)):
        s = -1 if PS.G_CONJ_COEF[g] else 1
        for k in range(6):
            g0, g1 = PS.G_CONSTS[g][k]
            for c in range(2):
                uv = (g0 % P, (-s * g1) % P) if c == 0 else (g1 % P, (s * g0) % P)
                if uv not in pairs:
                    pairs.append(uv)
                index[(g, k, c)] = pairs.index(uv)
    if (1, 0) not in pairs:
        pairs.append((1, 0))
    return pairs, index


GC_PAIRS, GC_INDEX = _frob_pairs()
GC_ONE = GC_PAIRS.index((1, 0))          # FINVC t = 0: the right-hand side 1


def row_program():
    rows = []

    def emit(typ, t=0, ra=-1, rb=-1, bsel=B_REG, gc=0, ld=-1, sk=SK_NONE, chm=-1, end=0):
        rows.append([typ, t, ra, rb, bsel, gc, ld, sk, chm, end])

    first_line = True
    for r in PS.SCHEDULE:
        gop, fop = r["gop"], r["fop"]
        if gop == PS.G_FQ:
            for t in range(8):
                emit(T_GFQ, t, end=END_FQ if t == 7 else 0)
        elif gop != PS.G_IDLE:
            sk = SK_TANGENT if gop == PS.G_TG else SK_CHORD
            chm = -1 if gop == PS.G_TG else gop - PS.G_CH0
            seq = [(T_GW, 0), (T_GW, 1), (T_GSL, 0), (T_GSL, 1), (T_GX3, 0), (T_GX3, 1), (T_GY3, 0), (T_GY3, 1), (T_GL1, 0), (T_GL1, 1),
                   (T_GL3, 0), (T_GL3, 1)]
            for i, (typ, c) in enumerate(seq):
                last = i == len(seq) - 1
                emit(typ, c, sk=sk, chm=chm, end=END_STEP if last and gop != PS.G_CH2 else 0)
        if fop == PS.F_IDLE:
            continue                      # (row 0 of the schedule loads 1 into the accumulator: here the first line is COPIED instead)
        if fop == PS.F_LINE and first_line:
            first_line = False
            for t in range(12):
                emit(T_FCOPY, t, bsel=B_LINE)
        elif fop in (PS.F_MUL, PS.F_LINE):
            for t in range(12):
                emit(T_FMUL, t, ra=r["ra"], rb=r["rb"] if fop == PS.F_MUL else -1, bsel=B_REG if fop == PS.F_MUL else B_LINE)
        elif fop == PS.F_FROB:
            for t in range(12):
                emit(T_FFROB, t, ra=r["ra"], gc=GC_INDEX[(r["gc"], t // 2, t % 2)])
        elif fop == PS.F_INV:
            for t in range(12):
                emit(T_FINVW, t, ra=r["ra"])
            for t in range(12):
                emit(T_FINVC, t, ra=r["ra"], bsel=B_CACC, gc=GC_ONE if t == 0 else 0)
        emit(T_FCOMMIT, ld=r["rd"])
    n_active = len(rows)
    assert n_active <= ROWS, n_active
    while len(rows) < ROWS:
        emit(T_IDLE)
    return rows, n_active


ROWPROG, N_ACTIVE = row_program()
RESULT_REG = PS.RESULT_REG
NREG = PS.NREG


# ---------------------------------------------------------------------------------------------------------------------------
def fq12_component(A, B, t):
    """component t = 2 k + c of A * B in Fq2[w] / (w^6 - xi), A, B as 12 Fq values (a_i = A[2i] + A[2i+1] u)"""
    k, c = divmod(t, 2)
    acc = 0
    for i in range(6):
        for j in range(6):
            a0, a1, b0, b1 = A[2 * i], A[2 * i + 1], B[2 * j], B[2 * j + 1]
            re, im = a0 * b0 - a1 * b1, a0 * b1 + a1 * b0
            if i + j == k:
                acc += re if c == 0 else im
            elif i + j == k + 6:                     # times xi = 9 + u
                acc += (9 * re - im) if c == 0 else (re + 9 * im)
    return acc % P


def simulate_rows(Pt, Q, trace=None):
    """runs ROWPROG on (P, Q); returns e(P, Q) as 12 MyFq12 coefficients.  Every check row's identity is asserted.  trace (a list)
    receives per row a dict of the state BEFORE the row's loads: regs, cacc, A, B, S, T, QS, Q1, Q2N, res, gc"""
    xp, yp = Pt
    regs = [[0] * 12 for _ in range(NREG)]
    cacc = [0] * 12
    S = [[0, 0] for _ in range(5)]
    T = [0, 0, 0, 0]                   # x.c0 x.c1 y.c0 y.c1
    Q1, Q2N = [0] * 4, [0] * 4
    Qf = [Q[0][0], Q[0][1], Q[1][0], Q[1][1]]
    FX, FY = bn.FROB_X, bn.FROB_Y
    f2m = bn.f2_mul
    for r, d in enumerate(ROWPROG):
        typ, t, ra, rb, bsel, gc, ld, sk, chm, end = d
        A = list(regs[ra]) if ra >= 0 else [0] * 12
        line = [yp % P, 0, S[3][0], S[3][1], 0, 0, S[4][0], S[4][1], 0, 0, 0, 0]
        B = [0] * 12
        if typ in (T_FMUL, T_FINVC, T_FCOPY):
            B = list(regs[rb]) if bsel == B_REG else line if bsel == B_LINE else list(cacc)
        QS = [0] * 4
        if sk == SK_CHORD:
            QS = list(Qf if chm == 0 else Q1 if chm == 1 else Q2N)
        u, v = GC_PAIRS[gc]
        res = 0
        tx, ty = (T[0], T[1]), (T[2], T[3])
        lam = (S[0][0], S[0][1])
        if typ == T_FMUL:
            res = fq12_component(A, B, t)
        elif typ == T_FFROB:
            k = t // 2
            res = (A[2 * k] * u + A[2 * k + 1] * v) % P
        elif typ == T_FINVW:
            if t == 0:
                inv = PS.t_inv([(A[2 * i], A[2 * i + 1]) for i in range(6)])
                simulate_rows._inv = [x for pr in inv for x in pr]
            res = simulate_rows._inv[t]
        elif typ == T_FINVC:
            assert fq12_component(A, B, t) == u, "inverse check row %d" % r
        elif typ == T_FCOPY:
            res = B[t]
        elif typ == T_GW:
            if t == 0:
                if sk == SK_TANGENT:
                    num, den = bn.f2_scal(f2m(tx, tx), 3), bn.f2_scal(ty, 2)
                else:
                    num, den = bn.f2_sub((QS[2], QS[3]), ty), bn.f2_sub((QS[0], QS[1]), tx)
                if den == (0, 0):
                    raise ValueError("degenerate step at row %d" % r)
                simulate_rows._lam = f2m(num, bn.f2_inv(den))
            res = simulate_rows._lam[t]
        elif typ == T_GSL:
            if sk == SK_TANGENT:
                lhs = bn.f2_sub(bn.f2_scal(f2m(lam, ty), 2), bn.f2_scal(f2m(tx, tx), 3))
            else:
                lhs = bn.f2_sub(f2m(lam, bn.f2_sub((QS[0], QS[1]), tx)), bn.f2_sub((QS[2], QS[3]), ty))
            assert lhs[t] == 0, "slope check row %d" % r
        elif typ == T_GX3:
            xb = tx if sk == SK_TANGENT else (QS[0], QS[1])
            res = bn.f2_sub(bn.f2_sub(f2m(lam, lam), tx), xb)[t]
        elif typ == T_GY3:
            res = bn.f2_sub(f2m(lam, bn.f2_sub(tx, (S[1][0], S[1][1]))), ty)[t]
        elif typ == T_GL1:
            res = bn.f2_neg(bn.f2_scal(lam, xp))[t]
        elif typ == T_GL3:
            res = bn.f2_sub(f2m(lam, tx), ty)[t]
        elif typ == T_GFQ:
            slot, c = divmod(t, 2)
            if slot == 0:
                val = f2m(bn.f2_conj((Qf[0], Qf[1])), FX)
            elif slot == 1:
                val = f2m(bn.f2_conj((Qf[2], Qf[3])), FY)
            elif slot == 2:
                val = f2m(bn.f2_conj((S[0][0], S[0][1])), FX)
            else:
                val = bn.f2_neg(f2m(bn.f2_conj((S[1][0], S[1][1])), FY))
            res = val[c]
        if trace is not None:
            trace.append(dict(regs=[list(x) for x in regs], cacc=list(cacc), A=A, B=B, S=[list(x) for x in S], T=list(T), QS=QS, Q1=list(Q1),
                              Q2N=list(Q2N), res=res, gc=(u, v)))
        # ---- end of row: loads
        if typ in (T_FMUL, T_FFROB, T_FINVW, T_FCOPY):
            cacc[t] = res
        elif typ == T_GW:
            S[0][t] = res
        elif typ in (T_GX3, T_GY3, T_GL1, T_GL3):
            S[{T_GX3: 1, T_GY3: 2, T_GL1: 3, T_GL3: 4}[typ]][t] = res
        elif typ == T_GFQ:
            S[t // 2][t % 2] = res
        if typ == T_FCOMMIT and ld >= 0:
            regs[ld] = list(cacc)
        if end == END_STEP:
            T = [S[1][0], S[1][1], S[2][0], S[2][1]]
        elif end == END_FQ:
            T = list(Qf)
            Q1 = [S[0][0], S[0][1], S[1][0], S[1][1]]
            Q2N = [S[2][0], S[2][1], S[3][0], S[3][1]]
    res12 = regs[RESULT_REG]
    return PS.t_to_c([(res12[2 * i], res12[2 * i + 1]) for i in range(6)])


if __name__ == "__main__":
    import random
    from collections import Counter
    print("active rows", N_ACTIVE, "of", ROWS, "| row constants", len(GC_PAIRS))
    print(Counter(TYPE_NAMES[d[0]] for d in ROWPROG))
    rnd = random.Random(11)
    s, t = rnd.randrange(1, bn.R), rnd.randrange(1, bn.R)
    Pt, Q = bn.g1_mul(bn.G1, s), bn.g2_mul(bn.G2, t)
    assert simulate_rows(Pt, Q) == bn.pairing(Pt, Q)
    assert simulate_rows(bn.G1, bn.G2) == bn.pairing(bn.G1, bn.G2)
    print("row program ok: simulate_rows == oracle/py/bn254.py::pairing")
