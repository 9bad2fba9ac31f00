/*
 * oracle/fri.h -- CPU restatement of plonky2's PolynomialBatch::prove_openings / fri_proof / verify_fri_proof for ARBITRARY
 * FriParams (rate_bits, cap height, reduction arities, proof-of-work rule, query count, salted oracles), fri.c.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (plonky2 @ InternetMaximalism/plonky2 541e127 is not vendored; the reference
 * reaches it through src/verifier_circuit.rs:133-135 -- the STARK sub-proofs -- and :225,253 -- the outer plonky2 proof).
 */
#ifndef ORACLE_FRI_H
#define ORACLE_FRI_H
#include "oracle.h"

#define ORC_POW_DUPLEX 0
#define ORC_POW_HASH 1
#define ORC_SALT_SIZE 4
#define ORC_FRI_MAX_ROUNDS 32

typedef struct {
    uint32_t rate_bits, cap_height, pow_bits, num_queries, pow_rule;
    uint32_t hiding;                            /* FriParams::hiding: salted oracles carry ORC_SALT_SIZE extra leaf words */
    uint32_t n_rounds;
    uint32_t arity_bits[ORC_FRI_MAX_ROUNDS];    /* FriParams::reduction_arity_bits */
} orc_fri_params;

/* FriReductionStrategy::ConstantArityBits(arity_bits, final_poly_bits) */
void orc_fri_const_arity(orc_fri_params *p, unsigned arity_bits, unsigned final_poly_bits, unsigned degree_bits);

/* a committed oracle: orc_batch (+ salt words at the end of every leaf when n_salt > 0) */
orc_batch *orc_batch_salted(const uint64_t *coeffs_or_values, int from_values, size_t ncols, unsigned log_n, unsigned rate_bits,
                            unsigned cap_height, const uint64_t *salt /* [n_salt][n << rate_bits], natural LDE order */,
                            size_t n_salt);

typedef struct { uint32_t oracle, col_begin, col_end; } orc_poly_range;
typedef struct {
    gl2 point;
    uint32_t n_ranges;
    const orc_poly_range *ranges;
} orc_fri_batch;   /* FriBatchInfo: a point and the polynomials opened there, as ranges of oracle columns */

/* growable word buffer the proof is appended to */
typedef struct { uint64_t *w; size_t len, cap; } orc_wbuf;
void orc_wb_push(orc_wbuf *b, const uint64_t *v, size_t n);

/* final polynomial of prove_openings: sum_i alpha^(k_i) (F_i(X) - F_i(z_i)) / (X - z_i), times X; n ext coefficients (malloc'ed) */
gl2 *orc_fri_final_poly(const orc_batch *const *oracles, const orc_fri_batch *batches, size_t n_batches, unsigned log_n, gl2 alpha);

/* commit phase + proof of work + query rounds, appended to `out` in the order
 *   commit caps [n_rounds][2^cap x 4] | final_poly[n >> sum(arity)] ext | pow_witness |
 *   queries[num_queries]: per oracle leaf[leaf_len] siblings[(log_m - cap) x 4]; per round evals[2^arity] ext, siblings */
int orc_fri_prove_core(const orc_batch *const *oracles, size_t n_oracles, unsigned log_n, const orc_fri_params *p,
                       const gl2 *final_coeffs, orc_challenger *ch, orc_wbuf *out, size_t *final_len);

/* the verifier side of the same section; opened[b][k] = claimed value of the k-th polynomial of batch b at its point.
 * caps[o], ncols[o] (polynomial columns), n_salt[o] describe the oracles.  Returns 0 or a negative code, *pos advanced. */
int orc_fri_verify_core(const uint64_t *proof, size_t len, size_t *pos, const uint64_t *const *caps, const int *ncols,
                        const int *n_salt, size_t n_oracles, const orc_fri_batch *batches, const gl2 *const *opened,
                        size_t n_batches, unsigned log_n, const orc_fri_params *p, gl2 alpha, orc_challenger *ch);

/* ---- generic opening proofs (the plonky2 outer prover's building block): flat layout
 *   header[8]: magic "SIPPFRI1", n_rounds, final_len, num_queries, n_oracles, n_batches, total_len, log_n
 *   opened values: for each batch, its polynomials' values at the point (ext)        (the caller's StarkOpeningSet / OpeningSet)
 *   then the core section above.
 * The challenger is the caller's transcript: the opened values are observed (batch by batch) before alpha is drawn. */
int orc_fri_prove_openings(const orc_batch *const *oracles, size_t n_oracles, const orc_fri_batch *batches, size_t n_batches,
                           unsigned log_n, const orc_fri_params *p, orc_challenger *ch, uint64_t **proof, size_t *len);
int orc_fri_verify_openings(const uint64_t *proof, size_t len, const uint64_t *const *caps, const int *ncols, const int *n_salt,
                            size_t n_oracles, const orc_fri_batch *batches, size_t n_batches, unsigned log_n,
                            const orc_fri_params *p, orc_challenger *ch);
#endif
