/*
 * oracle/bn254.h -- BN254 base-field arithmetic (Fq, Fq2, Fq12 in the w^12 - 18 w^6 + 82 form) for the
 * CPU restatement of the trace generators.  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (the reference uses
 * ark-bn254 0.4.0 and plonky2-bn254 @ d616d57 `MyFq12`, Cargo.toml:9,25, neither vendored).  Constants are
 * verified arithmetically; the Fq12 form is checked against the 2-3-2 tower by
 * oracle/py/bn254.py in tests/test_oracle_bn254.py.
 */
#ifndef ORACLE_BN254_H
#define ORACLE_BN254_H
#include <stdint.h>

typedef struct { uint64_t l[4]; } fq;       /* standard (non-Montgomery) representation, canonical */
typedef struct { fq c0, c1; } fq2;          /* c0 + c1 u, u^2 = -1 */
typedef struct { fq c[12]; } fq12;          /* sum c[k] w^k, w^12 = 18 w^6 - 82 */

void fq_init(void);
fq fq_from_u32(const uint32_t w[8]);
void fq_to_u32(fq a, uint32_t w[8]);
void fq_to_limbs16(fq a, uint16_t l[16]);
fq fq_from_limbs16(const uint16_t l[16]);
fq fq_zero(void);
fq fq_one(void);
fq fq_from_u64(uint64_t v);
int fq_is_zero(fq a);
int fq_eq(fq a, fq b);
fq fq_add(fq a, fq b);
fq fq_sub(fq a, fq b);
fq fq_neg(fq a);
fq fq_mul(fq a, fq b);
fq fq_inv(fq a);

fq2 fq2_add(fq2 a, fq2 b);
fq2 fq2_sub(fq2 a, fq2 b);
fq2 fq2_mul(fq2 a, fq2 b);
fq2 fq2_inv(fq2 a);
int fq2_is_zero(fq2 a);

fq12 fq12_mul(const fq12 *a, const fq12 *b);

#endif
