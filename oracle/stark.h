/*
 * oracle/stark.h -- CPU restatement of the STARK prover / verifier (see stark.c).  TEST INFRASTRUCTURE ONLY;
 * PARITY UNPINNED.
 */
#ifndef ORACLE_STARK_H
#define ORACLE_STARK_H
#include "air.h"

typedef struct {
    uint32_t rate_bits, cap_height, pow_bits, arity_bits, final_poly_bits, num_queries, num_challenges;
} orc_config;

void orc_default_config(orc_config *c);
/* returns 0 and a malloc'ed flat proof (free with orc_free) */
int orc_stark_prove(int kind, const uint32_t *ios, size_t num_io, const orc_config *cfg, uint64_t **proof_out,
                    size_t *proof_len);
/* 0 = accept; negative = the failing check (see stark.c) */
int orc_stark_verify(const uint64_t *proof, size_t len, const orc_config *cfg);
void orc_free(void *p);
#endif
