/*
 * oracle/mapg2.h -- the map Fp2 -> E'(Fp2) (RFC 9380 Shallue - van de Woestijne, Z = 1) with every intermediate value the
 * MapToG2 AIR holds.  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see mapg2.c).
 */
#ifndef ORACLE_MAPG2_H
#define ORACLE_MAPG2_H
#include "bn254.h"

/* checked witnesses in the column order of tools/air_gen.py::build_map_g2 */
enum { MG_T1, MG_TV1, MG_W, MG_TV3, MG_A4, MG_B4, MG_X2, MG_X1, MG_S1, MG_GX1, MG_S2, MG_GX2, MG_D, MG_E, MG_F, MG_X3, MG_S3,
       MG_GX3, MG_N1, MG_N2, MG_Y, MG_NWIT };

typedef struct { fq2 one, xi, b, c1, c2, c3, c4; } orc_mapg2_consts;
typedef struct {
    fq2 v[MG_NWIT];
    int e1, e2, z;
    fq2 m1, m2, xs, gxs;
} orc_mapg2_wit;

void orc_mapg2_constants(orc_mapg2_consts *k);
int orc_mapg2_witness(fq2 u, orc_mapg2_wit *w);
int orc_map_to_g2(const uint32_t *u, uint32_t *xy);
int orc_mapg2_record_sign_ok(const uint32_t *rec);
int fq2_sgn0(fq2 a);
int fq2_sqrt_even(fq2 a, fq2 *out);
#endif
