/*
 * oracle/air.h -- CPU restatement of the SIPP AIRs (trace + constraints).  TEST INFRASTRUCTURE ONLY;
 * PARITY UNPINNED (see air.c header).
 */
#ifndef ORACLE_AIR_H
#define ORACLE_AIR_H
#include "oracle.h"
#include "bn254.h"
#include "air_tables.h"

typedef struct {
    const air_spec_t *air;
    unsigned log_n;
    size_t num_io;   /* padded to a power of two */
    int width;       /* n_main + 2 n_checked */
    uint64_t *trace; /* [width][2^log_n], column major, natural row order */
    uint32_t *pis;   /* [num_io][pi_per_io] */
} orc_trace;

const air_spec_t *orc_air_get(int kind, unsigned log_n);   /* API kind 0 .. 6 */
/* kinds 4 / 5 (hardened G1 / G2) take the records of kinds 0 / 1 */
static inline int orc_record_kind(int kind) { return (kind == 4 || kind == 5) ? kind - 4 : kind; }
/* log2 of the trace rows per record: 512 (the exponentiations), 8 (MapToG2), 8192 (the final pairing: one identity per row) */
static inline unsigned orc_kind_log_rows(int kind) { return kind == 3 ? 3 : kind == 6 ? 13 : 9; }
int orc_air_api_kind(const air_spec_t *a);
int orc_air_width(const air_spec_t *a);
size_t orc_air_num_constraints(const air_spec_t *a);
orc_trace *orc_trace_build(int kind, const uint32_t *ios, size_t num_io, int *err);
int orc_records_on_curve(int kind, const uint32_t *pis, size_t num_io);
void orc_test_forge(int flags);   /* test hook, see air.c */
void orc_trace_free(orc_trace *t);
long orc_trace_check_row(const orc_trace *t, size_t row);

uint64_t orc_periodic_base(unsigned log_n, int which, uint64_t x);
gl2 orc_periodic_ext(unsigned log_n, int which, gl2 x);
/* value-periodic columns (air_spec_t::n_vflag + n_vconst of them, period 2^log_rows): number, interpolating coefficients
 * (2^log_rows of them: V_k(x) = P_k(x^(N / 2^log_rows))), evaluation at a point of the extension */
int orc_air_n_vper(const air_spec_t *a);
int orc_air_n_per(const air_spec_t *a);          /* AIR_N_PERIODIC + orc_air_n_vper */
void orc_vper_coeffs(const air_spec_t *a, int k, uint64_t *coeffs);
gl2 orc_vper_ext(const air_spec_t *a, unsigned log_n, const uint64_t *coeffs, gl2 x);
uint64_t orc_aux_value(const air_spec_t *a, const uint32_t *pis, size_t io, int ai);
void orc_aux_coeffs(const air_spec_t *a, const uint32_t *pis, size_t num_io, unsigned log_n, int ai, uint64_t *coeffs);

void orc_eval_base(const air_spec_t *air, const uint64_t *local, const uint64_t *next, const uint64_t *aux,
                   const uint64_t *per, const uint64_t *z_local, const uint64_t *z_next,
                   uint64_t lag_first, uint64_t lag_last, uint64_t z_last, const uint64_t alpha[2],
                   const uint64_t beta[2], const uint64_t gamma[2], uint64_t out[2]);
void orc_eval_ext(const air_spec_t *air, const gl2 *local, const gl2 *next, const gl2 *aux,
                  const gl2 *per, const gl2 *z_local, const gl2 *z_next, gl2 lag_first, gl2 lag_last,
                  gl2 z_last, const uint64_t alpha[2], const uint64_t beta[2], const uint64_t gamma[2], gl2 out[2]);
#endif
