/*
 * oracle/poseidon.c -- naive Poseidon-Goldilocks (width 12, x^7, 4+22+4 rounds).
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED beyond the three permutation KATs of
 * SURVEY.md App. E (upstream plonky2/src/hash/poseidon.rs @ 541e127 is absent).
 * Reference call sites: src/transcript_native.rs:27,57 (hash_n_to_hash_no_pad),
 * src/verifier_circuit.rs:196 (PoseidonGoldilocksConfig).
 */
#include "oracle.h"
#include "poseidon_constants.h"
#include <string.h>

/* the MDS entries are below 2^6: the two 32-bit halves of every state word are accumulated in plain 64-bit words (twelve
 * products of 32 x 6 bits each) and joined by ONE reduction per output -- the same value as sum_i s[(i + r) % 12] CIRC[i] + s[r] DIAG[r].
 * The loops run over the OUTPUT index innermost (contiguous, 32 x 32 -> 64 bit products) so that the compiler can use the vector
 * multiplier; two clones, picked by the loader for the machine the library runs on. */
__attribute__((target_clones("avx2", "default"), optimize("O3", "tree-vectorize")))
static void mds_layer(uint64_t s[12]) {
    uint32_t lo[24], hi[24];
    uint64_t al[12], ah[12];
    for (int i = 0; i < 12; i++) {
        lo[i] = lo[i + 12] = (uint32_t)s[i];
        hi[i] = hi[i + 12] = (uint32_t)(s[i] >> 32);
    }
    for (int r = 0; r < 12; r++) {
        al[r] = (uint64_t)lo[r] * (uint32_t)POSEIDON_DIAG[r];
        ah[r] = (uint64_t)hi[r] * (uint32_t)POSEIDON_DIAG[r];
    }
    for (int i = 0; i < 12; i++) {
        const uint32_t c = (uint32_t)POSEIDON_CIRC[i];
        for (int r = 0; r < 12; r++) {
            al[r] += (uint64_t)lo[i + r] * c;
            ah[r] += (uint64_t)hi[i + r] * c;
        }
    }
    for (int r = 0; r < 12; r++) s[r] = gl_reduce128((u128)al[r] + ((u128)ah[r] << 32));
}

static inline uint64_t sbox(uint64_t x) {
    uint64_t x2 = gl_sqr(x), x4 = gl_sqr(x2), x3 = gl_mul(x2, x);
    return gl_mul(x3, x4);
}

void orc_poseidon_permute(uint64_t s[12]) {
    int rnd = 0;
    for (int r = 0; r < 4; r++, rnd++) {
        for (int i = 0; i < 12; i++) s[i] = sbox(gl_add(s[i], POSEIDON_RC[12 * rnd + i]));
        mds_layer(s);
    }
    for (int r = 0; r < 22; r++, rnd++) {
        for (int i = 0; i < 12; i++) s[i] = gl_add(s[i], POSEIDON_RC[12 * rnd + i]);
        s[0] = sbox(s[0]);
        mds_layer(s);
    }
    for (int r = 0; r < 4; r++, rnd++) {
        for (int i = 0; i < 12; i++) s[i] = sbox(gl_add(s[i], POSEIDON_RC[12 * rnd + i]));
        mds_layer(s);
    }
}

void orc_hash_no_pad(const uint64_t *in, size_t n, uint64_t out[4]) {
    uint64_t s[12] = {0};
    for (size_t off = 0; off < n; off += 8) {
        size_t len = n - off < 8 ? n - off : 8;
        for (size_t i = 0; i < len; i++) s[i] = gl_canon(in[off + i]); /* overwrite mode */
        orc_poseidon_permute(s);
    }
    /* n == 0: plonky2 squeezes the zero state without permuting; never used on this path */
    memcpy(out, s, 4 * sizeof(uint64_t));
}

void orc_two_to_one(const uint64_t l[4], const uint64_t r[4], uint64_t out[4]) {
    uint64_t s[12] = {0};
    memcpy(s, l, 32);
    memcpy(s + 4, r, 32);
    orc_poseidon_permute(s);
    memcpy(out, s, 32);
}

void orc_hash_or_noop(const uint64_t *in, size_t n, uint64_t out[4]) {
    if (n <= 4) {
        memset(out, 0, 32);
        for (size_t i = 0; i < n; i++) out[i] = gl_canon(in[i]);
    } else {
        orc_hash_no_pad(in, n, out);
    }
}

/* ---------------- Challenger ---------------- */
void orc_chal_init(orc_challenger *c) { memset(c, 0, sizeof *c); }

static void duplexing(orc_challenger *c) {
    for (size_t i = 0; i < c->n_in; i++) c->state[i] = c->in_buf[i];
    c->n_in = 0;
    orc_poseidon_permute(c->state);
    memcpy(c->out_buf, c->state, 8 * sizeof(uint64_t));
    c->n_out = 8;
}

void orc_chal_observe(orc_challenger *c, uint64_t e) {
    c->n_out = 0;
    c->in_buf[c->n_in++] = gl_canon(e);
    if (c->n_in == 8) duplexing(c);
}
void orc_chal_observe_many(orc_challenger *c, const uint64_t *e, size_t n) {
    for (size_t i = 0; i < n; i++) orc_chal_observe(c, e[i]);
}
void orc_chal_observe_ext(orc_challenger *c, gl2 e) {
    orc_chal_observe(c, e.c0);
    orc_chal_observe(c, e.c1);
}
void orc_chal_observe_cap(orc_challenger *c, const uint64_t *cap, size_t n_digests) {
    orc_chal_observe_many(c, cap, 4 * n_digests);
}
uint64_t orc_chal_get(orc_challenger *c) {
    if (c->n_in != 0 || c->n_out == 0) duplexing(c);
    return c->out_buf[--c->n_out]; /* pops from the END */
}
gl2 orc_chal_get_ext(orc_challenger *c) {
    gl2 r;
    r.c0 = orc_chal_get(c);
    r.c1 = orc_chal_get(c);
    return r;
}
