/*
 * oracle/pairing.h -- the final-pairing AIR's primary witness on the CPU (see pairing.c).  TEST INFRASTRUCTURE ONLY;
 * PARITY UNPINNED.
 */
#ifndef ORACLE_PAIRING_H
#define ORACLE_PAIRING_H
#include <stddef.h>
#include "bn254.h"
#include "air_tables.h"

/* record = P (16 u32) | Q (32 u32) | Z (96 u32).  1 when P is on E(Fp), Q on E'(Fp2) and [r] Q = O */
int orc_pairing_record_ok(const uint32_t *rec);
int orc_pairing_run(const air_spec_t *a, uint64_t *tr, size_t n, size_t row0, const uint32_t *rec, uint32_t *out_words);
int orc_pairing(const uint32_t *pq, uint32_t *out96);
#endif
