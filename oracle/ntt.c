/*
 * oracle/ntt.c -- Goldilocks FFT / iFFT / coset LDE with plonky2's conventions
 * (upstream field/src/fft.rs, polynomial/mod.rs @ 541e127,
 * absent from /root/reference).  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED --
 * pinned here by the identity fft == naive DFT (tests/test_oracle_ntt.py).
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

void orc_naive_dft(const uint64_t *in, uint64_t *out, unsigned log_n) {
    size_t n = (size_t)1 << log_n;
    uint64_t w = gl_root_of_unity(log_n);
    for (size_t i = 0; i < n; i++) {
        uint64_t wi = gl_pow(w, i), x = 1, acc = 0;
        for (size_t j = 0; j < n; j++) {
            acc = gl_add(acc, gl_mul(in[j], x));
            x = gl_mul(x, wi);
        }
        out[i] = acc;
    }
}

/* iterative radix-2: bit-reverse then DIT; natural in, natural out */
static void fft_core(uint64_t *a, unsigned log_n, uint64_t root) {
    size_t n = (size_t)1 << log_n;
    for (size_t i = 0; i < n; i++) {
        size_t j = bitrev32((uint32_t)i, log_n);
        if (i < j) { uint64_t t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    for (unsigned s = 1; s <= log_n; s++) {
        size_t m = (size_t)1 << s, h = m >> 1;
        uint64_t wm = root;
        for (unsigned k = s; k < log_n; k++) wm = gl_sqr(wm);
        uint64_t *tw = (uint64_t *)malloc(h * sizeof(uint64_t));
        tw[0] = 1;
        for (size_t j = 1; j < h; j++) tw[j] = gl_mul(tw[j - 1], wm);
        for (size_t k = 0; k < n; k += m)
            for (size_t j = 0; j < h; j++) {
                uint64_t t = gl_mul(tw[j], a[k + j + h]);
                uint64_t u = a[k + j];
                a[k + j] = gl_add(u, t);
                a[k + j + h] = gl_sub(u, t);
            }
        free(tw);
    }
}

void orc_fft(uint64_t *a, unsigned log_n) { fft_core(a, log_n, gl_root_of_unity(log_n)); }

void orc_ifft(uint64_t *a, unsigned log_n) {
    size_t n = (size_t)1 << log_n;
    fft_core(a, log_n, gl_inv(gl_root_of_unity(log_n)));
    uint64_t ninv = gl_inv((uint64_t)n % GL_P);
    for (size_t i = 0; i < n; i++) a[i] = gl_mul(a[i], ninv);
}

void orc_coset_lde(const uint64_t *coeffs, unsigned log_n, unsigned rate_bits, uint64_t shift, uint64_t *out) {
    size_t n = (size_t)1 << log_n, m = n << rate_bits;
    uint64_t s = 1;
    for (size_t i = 0; i < n; i++) { out[i] = gl_mul(coeffs[i], s); s = gl_mul(s, shift); }
    memset(out + n, 0, (m - n) * sizeof(uint64_t));
    orc_fft(out, log_n + rate_bits);
}

/* extension-field transforms: componentwise (the twiddles are base-field) */
void orc_fft_ext(gl2 *a, unsigned log_n) {
    size_t n = (size_t)1 << log_n;
    uint64_t *t = (uint64_t *)malloc(n * sizeof(uint64_t));
    for (int c = 0; c < 2; c++) {
        for (size_t i = 0; i < n; i++) t[i] = c ? a[i].c1 : a[i].c0;
        orc_fft(t, log_n);
        for (size_t i = 0; i < n; i++) { if (c) a[i].c1 = t[i]; else a[i].c0 = t[i]; }
    }
    free(t);
}
void orc_ifft_ext(gl2 *a, unsigned log_n) {
    size_t n = (size_t)1 << log_n;
    uint64_t *t = (uint64_t *)malloc(n * sizeof(uint64_t));
    for (int c = 0; c < 2; c++) {
        for (size_t i = 0; i < n; i++) t[i] = c ? a[i].c1 : a[i].c0;
        orc_ifft(t, log_n);
        for (size_t i = 0; i < n; i++) { if (c) a[i].c1 = t[i]; else a[i].c0 = t[i]; }
    }
    free(t);
}
