/*
 * oracle/merkle.c -- Poseidon Merkle tree with cap + PolynomialBatch commit,
 * restating plonky2 hash/merkle_tree.rs and fri/oracle.rs @ 541e127 (absent
 * from /root/reference).  TEST INFRASTRUCTURE ONLY;
 * PARITY UNPINNED.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
int orc_get_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* `own` != 0: the tree takes `leaves` (malloc'ed by the caller) over instead of copying it -- the batches of the large configurations
 * (n = 1024: 12.5 GB of leaves per oracle) would otherwise hold two copies at their peak */
static orc_merkle *merkle_build(uint64_t *leaves, int own, unsigned log_leaves, size_t leaf_len, unsigned cap_height) {
    orc_merkle *t = (orc_merkle *)calloc(1, sizeof *t);
    size_t n = (size_t)1 << log_leaves;
    if (cap_height > log_leaves) cap_height = log_leaves;
    t->log_leaves = log_leaves;
    t->cap_height = cap_height;
    t->leaf_len = leaf_len;
    if (own) t->leaves = leaves;
    else {
        t->leaves = (uint64_t *)malloc(n * leaf_len * sizeof(uint64_t));
        memcpy(t->leaves, leaves, n * leaf_len * sizeof(uint64_t));
    }
    unsigned n_levels = log_leaves - cap_height + 1;
    t->level_off = (size_t *)malloc((n_levels + 1) * sizeof(size_t));
    size_t total = 0;
    for (unsigned l = 0; l < n_levels; l++) { t->level_off[l] = total; total += n >> l; }
    t->level_off[n_levels] = total;
    t->digests = (uint64_t *)malloc(total * 4 * sizeof(uint64_t));
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) orc_hash_or_noop(t->leaves + i * leaf_len, leaf_len, t->digests + 4 * i);
    for (unsigned l = 1; l < n_levels; l++) {
        uint64_t *prev = t->digests + 4 * t->level_off[l - 1];
        uint64_t *cur = t->digests + 4 * t->level_off[l];
        size_t cnt = n >> l;
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < cnt; i++) orc_two_to_one(prev + 8 * i, prev + 8 * i + 4, cur + 4 * i);
    }
    t->cap = t->digests + 4 * t->level_off[n_levels - 1];
    return t;
}
orc_merkle *orc_merkle_new(const uint64_t *leaves, unsigned log_leaves, size_t leaf_len, unsigned cap_height) {
    return merkle_build((uint64_t *)leaves, 0, log_leaves, leaf_len, cap_height);
}

void orc_merkle_free(orc_merkle *t) {
    if (!t) return;
    free(t->leaves); free(t->digests); free(t->level_off); free(t);
}

size_t orc_merkle_prove(const orc_merkle *t, size_t index, uint64_t *siblings) {
    unsigned n_sib = t->log_leaves - t->cap_height;
    for (unsigned l = 0; l < n_sib; l++) {
        size_t sib = (index >> l) ^ 1;
        memcpy(siblings + 4 * l, t->digests + 4 * (t->level_off[l] + sib), 32);
    }
    return n_sib;
}

int orc_merkle_verify(const uint64_t *leaf, size_t leaf_len, size_t index, const uint64_t *siblings,
                      size_t n_siblings, const uint64_t *cap, unsigned cap_height) {
    uint64_t cur[4];
    (void)cap_height;
    orc_hash_or_noop(leaf, leaf_len, cur);
    for (size_t l = 0; l < n_siblings; l++) {
        uint64_t nxt[4];
        if ((index >> l) & 1) orc_two_to_one(siblings + 4 * l, cur, nxt);
        else orc_two_to_one(cur, siblings + 4 * l, nxt);
        memcpy(cur, nxt, 32);
    }
    return memcmp(cur, cap + 4 * (index >> n_siblings), 32) == 0;
}

orc_batch *orc_batch_from_coeffs(const uint64_t *coeffs, size_t ncols, unsigned log_n, unsigned rate_bits,
                                 unsigned cap_height) {
    orc_batch *b = (orc_batch *)calloc(1, sizeof *b);
    size_t n = (size_t)1 << log_n, m = n << rate_bits;
    unsigned log_m = log_n + rate_bits;
    b->log_n = log_n; b->rate_bits = rate_bits; b->cap_height = cap_height; b->ncols = ncols;
    b->coeffs = (uint64_t *)malloc(ncols * n * sizeof(uint64_t));
    memcpy(b->coeffs, coeffs, ncols * n * sizeof(uint64_t));
    uint64_t *leaves = (uint64_t *)malloc(m * ncols * sizeof(uint64_t));
#pragma omp parallel
    {
        uint64_t *col = (uint64_t *)malloc(m * sizeof(uint64_t));
#pragma omp for schedule(dynamic)
        for (size_t c = 0; c < ncols; c++) {
            orc_coset_lde(b->coeffs + c * n, log_n, rate_bits, GL_GEN, col);
            /* transpose + reverse_index_bits: leaf j holds natural row bitrev(j) */
            for (size_t j = 0; j < m; j++) leaves[j * ncols + c] = col[bitrev32((uint32_t)j, log_m)];
        }
        free(col);
    }
    b->tree = merkle_build(leaves, 1, log_m, ncols, cap_height);
    return b;
}

orc_batch *orc_batch_from_values(const uint64_t *values, size_t ncols, unsigned log_n, unsigned rate_bits,
                                 unsigned cap_height) {
    size_t n = (size_t)1 << log_n;
    uint64_t *coeffs = (uint64_t *)malloc(ncols * n * sizeof(uint64_t));
    memcpy(coeffs, values, ncols * n * sizeof(uint64_t));
#pragma omp parallel for schedule(dynamic)
    for (size_t c = 0; c < ncols; c++) orc_ifft(coeffs + c * n, log_n);
    orc_batch *b = orc_batch_from_coeffs(coeffs, ncols, log_n, rate_bits, cap_height);
    free(coeffs);
    return b;
}

void orc_batch_free(orc_batch *b) {
    if (!b) return;
    free(b->coeffs); orc_merkle_free(b->tree); free(b);
}
const uint64_t *orc_batch_cap(const orc_batch *b) { return b->tree->cap; }
const uint64_t *orc_batch_leaves(const orc_batch *b) { return b->tree->leaves; }
const uint64_t *orc_batch_coeffs(const orc_batch *b) { return b->coeffs; }
const uint64_t *orc_batch_level(const orc_batch *b, unsigned level, size_t *n_digests) {
    if (n_digests) *n_digests = ((size_t)1 << b->tree->log_leaves) >> level;
    return b->tree->digests + 4 * b->tree->level_off[level];
}
