/* sipp_amd/csrc/pairing_rowsrc.h -- WHERE every primary cell of the final-pairing AIR's trace comes from (API kind 6;
 * tools/air_gen.py::build_pairing, row program tools/pairing_rows.py; reference src/bin/bls_aggregation.rs:76-77).
 *
 * The row program is data (AIR_PAIRING_ROWPROG): a row produces at most ONE new field element (its RES) and every other primary cell
 * holds a value produced earlier (a register, an operand, the accumulator ...), a constant, a coordinate of the record or zero.  This
 * header replays the program SYMBOLICALLY on the host: for each of the 8192 rows and each of the 147 field elements of a row it
 * returns the index of the value in a per-record POOL
 *     [0] zero | [1..2] P | [3..6] Q | [7..10] Frobenius constants of the twist | [11 ..] the row constants (u, v) | [PP_ROW + r] RES of row r
 * so that the GPU computes the ~6000 values of a pairing once (one wave per record walking the operation schedule) and a second,
 * fully parallel kernel writes the 2352 limb cells of all rows from the pool.  Plain C: the same function is compiled into a CPU test
 * (tests/test_host_pairing_rowsrc.py) that rebuilds the oracle's trace from the oracle's own RES column through this table. */
#ifndef SIPP_PAIRING_ROWSRC_H
#define SIPP_PAIRING_ROWSRC_H
#include <stdint.h>
#include <string.h>

#include "air_tables.h"

enum { PP_ZERO = 0, PP_PX = 1, PP_PY = 2, PP_Q = 3, PP_FROBC = 7, PP_GC = 11, PP_ROW = PP_GC + 2 * AIR_PAIRING_NGC + 3, PP_N = PP_ROW + AIR_PAIRING_ROWS };
/* field elements of a row in layout order (AIR_PAIRING_LAYOUT_*): PX PY | QX QY Q1X Q1Y Q2X Q2Y TX TY QSX QSY FXC FYC (2 each) | SR (10) |
 * GC (2) | A B CACC (12 each) | REG (12 per register) | RES */
enum { PE_PX = 0, PE_PY = 1, PE_PTS = 2, PE_SR = 26, PE_GC = 36, PE_A = 38, PE_B = 50, PE_CACC = 62, PE_REG = 74, PE_RES = PE_REG + 12 * AIR_PAIRING_NREG,
       PE_N = PE_RES + 1 };
/* row types and descriptor fields of tools/pairing_rows.py */
enum { PRT_IDLE, PRT_FMUL, PRT_FFROB, PRT_FINVW, PRT_FINVC, PRT_FCOPY, PRT_FCOMMIT, PRT_GW, PRT_GSL, PRT_GX3, PRT_GY3, PRT_GL1, PRT_GL3, PRT_GFQ };
enum { PRF_TYP, PRF_T, PRF_RA, PRF_RB, PRF_BSEL, PRF_GC, PRF_LD, PRF_SK, PRF_CHM, PRF_END };

/* first trace column of element e (16 cells; RES: 16 * cells_per_limb) */
static inline int pairing_elem_col(const int32_t *lay, int e) {
    if (e < PE_PTS) return lay[e];
    if (e < PE_SR) return lay[2 + (e - PE_PTS) / 2] + 16 * ((e - PE_PTS) & 1);
    if (e < PE_GC) return lay[14] + 16 * (e - PE_SR);
    if (e < PE_A) return lay[15] + 16 * (e - PE_GC);
    if (e < PE_B) return lay[16] + 16 * (e - PE_A);
    if (e < PE_CACC) return lay[17] + 16 * (e - PE_B);
    if (e < PE_REG) return lay[18] + 16 * (e - PE_CACC);
    if (e < PE_RES) return lay[19] + 16 * (e - PE_REG);
    return lay[20];
}

/* src[e * AIR_PAIRING_ROWS + r] = pool index of element e on row r.  Returns 0, or -1 when the program reads something this replay does
 * not know (a register index out of range). */
static inline int pairing_row_sources(uint16_t *src) {
    enum { NR = AIR_PAIRING_NREG, R = AIR_PAIRING_ROWS };
    uint16_t regs[NR][12], cacc[12], S[5][2], T[4], Q1[4], Q2[4];
    memset(regs, 0, sizeof regs);
    memset(cacc, 0, sizeof cacc);
    memset(S, 0, sizeof S);
    memset(T, 0, sizeof T);
    memset(Q1, 0, sizeof Q1);
    memset(Q2, 0, sizeof Q2);
    for (int r = 0; r < R; r++) {
        const int8_t *d = AIR_PAIRING_ROWPROG[r];
        const int typ = d[PRF_TYP], t = d[PRF_T], ra = d[PRF_RA], rb = d[PRF_RB], bsel = d[PRF_BSEL], gc = d[PRF_GC], ld = d[PRF_LD],
                  sk = d[PRF_SK], chm = d[PRF_CHM], end = d[PRF_END];
        if (ra >= NR || rb >= NR || ld >= NR || gc < 0 || gc >= AIR_PAIRING_NGC || t < 0 || t >= 12) return -1;
        uint16_t A[12], B[12], QS[4] = {0, 0, 0, 0};
        for (int i = 0; i < 12; i++) {
            A[i] = ra >= 0 ? regs[ra][i] : (uint16_t)PP_ZERO;
            B[i] = PP_ZERO;
        }
        if (rb >= 0) {
            for (int i = 0; i < 12; i++) B[i] = regs[rb][i];
        } else if (bsel == 1) {
            B[0] = PP_PY; B[2] = S[3][0]; B[3] = S[3][1]; B[6] = S[4][0]; B[7] = S[4][1];
        } else if (bsel == 2) {
            for (int i = 0; i < 12; i++) B[i] = cacc[i];
        }
        if (sk == 2)
            for (int i = 0; i < 4; i++) QS[i] = chm == 0 ? (uint16_t)(PP_Q + i) : chm == 1 ? Q1[i] : Q2[i];
        uint16_t res = PP_ZERO;
        switch (typ) {
            case PRT_FMUL: case PRT_FFROB: case PRT_FINVW: case PRT_GW: case PRT_GX3: case PRT_GY3: case PRT_GL1: case PRT_GL3: case PRT_GFQ:
                res = (uint16_t)(PP_ROW + r);
                break;
            case PRT_FCOPY:
                res = B[t];
                break;
            default:
                break;
        }
        uint16_t *o = src + r;
#define PUT(e, v) o[(size_t)(e) * R] = (v)
        PUT(PE_PX, PP_PX);
        PUT(PE_PY, PP_PY);
        for (int i = 0; i < 4; i++) {
            PUT(PE_PTS + i, PP_Q + i);
            PUT(PE_PTS + 4 + i, Q1[i]);
            PUT(PE_PTS + 8 + i, Q2[i]);
            PUT(PE_PTS + 12 + i, T[i]);
            PUT(PE_PTS + 16 + i, QS[i]);
            PUT(PE_PTS + 20 + i, PP_FROBC + i);
        }
        for (int s = 0; s < 5; s++) {
            PUT(PE_SR + 2 * s, S[s][0]);
            PUT(PE_SR + 2 * s + 1, S[s][1]);
        }
        PUT(PE_GC, PP_GC + 2 * gc);
        PUT(PE_GC + 1, PP_GC + 2 * gc + 1);
        for (int i = 0; i < 12; i++) {
            PUT(PE_A + i, A[i]);
            PUT(PE_B + i, B[i]);
            PUT(PE_CACC + i, cacc[i]);
            for (int k = 0; k < NR; k++) PUT(PE_REG + 12 * k + i, regs[k][i]);
        }
        PUT(PE_RES, res);
#undef PUT
        /* end of the row: loads (tools/pairing_rows.py::simulate_rows) */
        if (typ == PRT_FMUL || typ == PRT_FFROB || typ == PRT_FINVW || typ == PRT_FCOPY) cacc[t] = res;
        else if (typ == PRT_GW) S[0][t & 1] = res;
        else if (typ == PRT_GX3) S[1][t & 1] = res;
        else if (typ == PRT_GY3) S[2][t & 1] = res;
        else if (typ == PRT_GL1) S[3][t & 1] = res;
        else if (typ == PRT_GL3) S[4][t & 1] = res;
        else if (typ == PRT_GFQ) S[(t >> 1) & 3][t & 1] = res;
        if (typ == PRT_FCOMMIT && ld >= 0)
            for (int i = 0; i < 12; i++) regs[ld][i] = cacc[i];
        if (end == 1) {
            T[0] = S[1][0]; T[1] = S[1][1]; T[2] = S[2][0]; T[3] = S[2][1];
        } else if (end == 2) {
            for (int i = 0; i < 4; i++) T[i] = (uint16_t)(PP_Q + i);
            Q1[0] = S[0][0]; Q1[1] = S[0][1]; Q1[2] = S[1][0]; Q1[3] = S[1][1];
            Q2[0] = S[2][0]; Q2[1] = S[2][1]; Q2[2] = S[3][0]; Q2[3] = S[3][1];
        }
    }
    return 0;
}

/* where the value walk (one wave per record over the OPERATION schedule AIR_PAIRING_SCHED) logs its results: oprow[t] = the first of
 * the twelve result rows of operation t (-1: none / not logged: idle operations, and the first line, which the row program copies),
 * steprow[s] = the first of the twelve rows of point step s (slope at +0, x3 at +4, y3 at +6, -lam x_P at +8, lam x_T - y_T at +10);
 * the Frobenius images of Q are the results of rows 0 .. 7.  Returns the number of point steps, or -1 when the row program and the
 * operation schedule do not pair up. */
static inline int pairing_log_rows(int16_t *oprow, int16_t *steprow, int max_steps) {
    int r = 0, steps = 0;
    if (AIR_PAIRING_SCHED[0][5] != 5 || AIR_PAIRING_ROWPROG[0][PRF_TYP] != PRT_GFQ) return -1;
    for (int t = 0; t < AIR_PAIRING_OPS; t++) {
        const int fop = AIR_PAIRING_SCHED[t][0], gop = AIR_PAIRING_SCHED[t][5];
        oprow[t] = -1;
        if (gop == 5) {
            for (int i = 0; i < 8; i++, r++)
                if (r >= AIR_PAIRING_ROWS || AIR_PAIRING_ROWPROG[r][PRF_TYP] != PRT_GFQ || AIR_PAIRING_ROWPROG[r][PRF_T] != i) return -1;
        } else if (gop >= 1 && gop <= 4) {
            static const int8_t seq[12] = {PRT_GW, PRT_GW, PRT_GSL, PRT_GSL, PRT_GX3, PRT_GX3, PRT_GY3, PRT_GY3, PRT_GL1, PRT_GL1, PRT_GL3, PRT_GL3};
            if (steps >= max_steps) return -1;
            steprow[steps++] = (int16_t)r;
            for (int i = 0; i < 12; i++, r++)
                if (r >= AIR_PAIRING_ROWS || AIR_PAIRING_ROWPROG[r][PRF_TYP] != seq[i] || AIR_PAIRING_ROWPROG[r][PRF_T] != (i & 1)) return -1;
        }
        if (fop == 0) continue;
        if (r + 13 > AIR_PAIRING_ROWS) return -1;
        const int typ = AIR_PAIRING_ROWPROG[r][PRF_TYP];
        const int want = fop == 3 ? PRT_FINVW : fop == 4 ? PRT_FFROB : PRT_FMUL;
        if (typ == PRT_FCOPY && fop == 2) {
            r += 12;
        } else {
            if (typ != want || AIR_PAIRING_ROWPROG[r][PRF_T] != 0) return -1;
            oprow[t] = (int16_t)r;
            r += fop == 3 ? 24 : 12;
        }
        if (r >= AIR_PAIRING_ROWS || AIR_PAIRING_ROWPROG[r][PRF_TYP] != PRT_FCOMMIT || AIR_PAIRING_ROWPROG[r][PRF_LD] != AIR_PAIRING_SCHED[t][4]) return -1;
        r++;
    }
    if (r != AIR_PAIRING_ACTIVE_ROWS) return -1;
    return steps;
}
#endif
