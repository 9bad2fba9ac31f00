/*
 * oracle/stark.h -- CPU restatement of the STARK prover / verifier (see stark.c).  TEST INFRASTRUCTURE ONLY;
 * PARITY UNPINNED.
 */
#ifndef ORACLE_STARK_H
#define ORACLE_STARK_H
#include "air.h"

typedef struct {
    uint32_t rate_bits, cap_height, pow_bits, arity_bits, final_poly_bits, num_queries, num_challenges;
    uint32_t pow_rule; /* ORC_POW_DUPLEX (default) or ORC_POW_HASH: which recollection of upstream's grind rule (stark.c) */
    uint32_t fs_rule;     /* 0 (default): the statement is observed before the trace cap (this repository's format); 1: starky's
                             recalled order -- the challenger starts at the trace cap */
    uint32_t lookup_rule; /* 0 (default): independent (beta, gamma) for a lookup's two permutation factors; 1: both factors under
                             gamma (starky's single-column permutation pairs as recalled; beta drawn and unused) */
} orc_config;
#define ORC_POW_DUPLEX 0
#define ORC_POW_HASH 1

void orc_default_config(orc_config *c);
/* returns 0 and a malloc'ed flat proof (free with orc_free) */
int orc_stark_prove(int kind, const uint32_t *ios, size_t num_io, const orc_config *cfg, uint64_t **proof_out,
                    size_t *proof_len);
/* the same from a filled (possibly tampered) trace; the trace stays the caller's */
int orc_stark_prove_trace(const orc_trace *t, const orc_config *cfg, uint64_t **proof_out, size_t *proof_len);
/* test hook (one shot): add `delta` to cell (col, row) of the Z values (stage 1) or the quotient chunk coefficients (stage 2)
 * inside the next orc_stark_prove_trace */
void orc_test_tamper(int stage, int col, size_t row, uint64_t delta);
/* statement binding helpers (shared with the tests) */
int orc_pis_canonical(int kind, const uint32_t *pis, size_t num_io);
void orc_pi_root(const uint32_t *pis, size_t num_io, int ppi, uint64_t root[4]);
/* 0 = accept; negative = the failing check (see stark.c) */
int orc_stark_verify(const uint64_t *proof, size_t len, const orc_config *cfg);
void orc_free(void *p);
#endif
