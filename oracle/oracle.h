/*
 * oracle/oracle.h -- public interface of the CPU restatement (liboracle.so).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/gl.h header).  PARITY UNPINNED: the
 * algorithms restated here live in plonky2/starky @ InternetMaximalism/plonky2
 * rev 541e127 and starky-bn254 @ 2d46f9e (reference Cargo.toml:21-27), none of
 * which is under /root/reference.  What pins them here: SURVEY.md App. A
 * (recalled upstream behaviour), App. E (Poseidon constants + 3 permutation
 * KATs, reproduced by tests/test_oracle_poseidon.py) and mathematical
 * identities (NTT == naive DFT, LDE == polynomial evaluation, STARK
 * self-verification).
 */
#ifndef ORACLE_H
#define ORACLE_H

#include "gl.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------- Poseidon (plonky2 hash/poseidon.rs, hashing.rs) ------------- */
void orc_poseidon_permute(uint64_t state[12]);
/* hash_n_to_hash_no_pad: overwrite-mode sponge, rate 8, no padding.  Used by the
 * reference at src/transcript_native.rs:27,57. */
void orc_hash_no_pad(const uint64_t *in, size_t n, uint64_t out[4]);
void orc_two_to_one(const uint64_t l[4], const uint64_t r[4], uint64_t out[4]);
/* OpenMP team size of every parallel region of the oracle.  The environment variable OMP_NUM_THREADS is read when the runtime is
 * first loaded -- in a Python process that has imported torch that was long ago -- so a caller that wants a thread count sets it
 * here; orc_get_max_threads() is what the next parallel region will really use (1 in a build without OpenMP). */
void orc_set_num_threads(int n);
int orc_get_max_threads(void);
/* hash_or_noop: len <= 4 -> zero padded elements ARE the digest */
void orc_hash_or_noop(const uint64_t *in, size_t n, uint64_t out[4]);

/* ---------------- NTT (plonky2 field/src/fft.rs conventions) ------------------ */
/* natural in, natural out: out[i] = sum_j c[j] w^(i j) */
void orc_fft(uint64_t *a, unsigned log_n);
void orc_ifft(uint64_t *a, unsigned log_n);
void orc_naive_dft(const uint64_t *in, uint64_t *out, unsigned log_n);
/* coset LDE of coefficient vector (len 2^log_n) to 2^(log_n+rate_bits) points,
 * natural order: out[i] = f(shift * w_{2N}^i) */
void orc_coset_lde(const uint64_t *coeffs, unsigned log_n, unsigned rate_bits, uint64_t shift, uint64_t *out);
void orc_fft_ext(gl2 *a, unsigned log_n);
void orc_ifft_ext(gl2 *a, unsigned log_n);

/* ---------------- Merkle tree (plonky2 hash/merkle_tree.rs) ------------------- */
typedef struct {
    unsigned log_leaves;
    unsigned cap_height;
    size_t leaf_len;
    uint64_t *leaves;  /* n_leaves * leaf_len, row major (owned) */
    uint64_t *digests; /* level 0 (leaf digests) .. level (log_leaves - cap_height); each 4 u64 */
    size_t *level_off; /* offset (in digests of 4) of each level */
    uint64_t *cap;     /* 2^cap_height * 4 */
} orc_merkle;

/* takes ownership of nothing: copies `leaves` */
orc_merkle *orc_merkle_new(const uint64_t *leaves, unsigned log_leaves, size_t leaf_len, unsigned cap_height);
void orc_merkle_free(orc_merkle *t);
/* siblings from leaf level up to (not including) the cap level; returns count */
size_t orc_merkle_prove(const orc_merkle *t, size_t index, uint64_t *siblings /* (log_leaves-cap_height)*4 */);
int orc_merkle_verify(const uint64_t *leaf, size_t leaf_len, size_t index, const uint64_t *siblings,
                      size_t n_siblings, const uint64_t *cap, unsigned cap_height);

/* flat helpers for tests: commit a column-major batch the way
 * PolynomialBatch::from_values does (iFFT, coset LDE shift 7 blowup 2^rate_bits,
 * transpose, bit-reverse rows, Merkle).  `values` is [ncols][2^log_n].
 * Outputs: coeffs [ncols][N] (natural), lde_leaves [2N][ncols] in leaf order
 * (leaf j = natural LDE row bitrev(j)), digests of every level, cap. */
typedef struct {
    unsigned log_n, rate_bits, cap_height;
    size_t ncols;
    uint64_t *coeffs;   /* [ncols][N] */
    orc_merkle *tree;   /* leaves = [2^(log_n+rate_bits)][ncols] */
} orc_batch;

orc_batch *orc_batch_from_values(const uint64_t *values, size_t ncols, unsigned log_n, unsigned rate_bits,
                                 unsigned cap_height);
orc_batch *orc_batch_from_coeffs(const uint64_t *coeffs, size_t ncols, unsigned log_n, unsigned rate_bits,
                                 unsigned cap_height);
void orc_batch_free(orc_batch *b);
const uint64_t *orc_batch_cap(const orc_batch *b);
const uint64_t *orc_batch_leaves(const orc_batch *b);
const uint64_t *orc_batch_coeffs(const orc_batch *b);
const uint64_t *orc_batch_level(const orc_batch *b, unsigned level, size_t *n_digests);

/* ---------------- Challenger (plonky2 iop/challenger.rs) ---------------------- */
typedef struct {
    uint64_t state[12];
    uint64_t in_buf[8];
    size_t n_in;
    uint64_t out_buf[8];
    size_t n_out;
} orc_challenger;

void orc_chal_init(orc_challenger *c);
void orc_chal_observe(orc_challenger *c, uint64_t e);
void orc_chal_observe_many(orc_challenger *c, const uint64_t *e, size_t n);
void orc_chal_observe_ext(orc_challenger *c, gl2 e);
void orc_chal_observe_cap(orc_challenger *c, const uint64_t *cap, size_t n_digests);
uint64_t orc_chal_get(orc_challenger *c);
gl2 orc_chal_get_ext(orc_challenger *c);

#ifdef __cplusplus
}
#endif
#endif
