/*
 * oracle/plonk_witness.c -- the witness generators of the gate families a plonky2 recursive verifier is made of, row by row, in plain C.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED: restated from the published structure of plonky2 @ 541e127 (un-vendored): iop/generator.rs
 * generate_partial_witness -- what `data.prove(pw)` at reference src/verifier_circuit.rs:253 runs before the first commitment -- and the
 * generators of gates/arithmetic_base.rs, base_sum.rs (BaseSplitGenerator), constant.rs, public_input.rs, random_access.rs, reducing.rs,
 * poseidon.rs and plonky2_u32's arithmetic_u32.rs.  The wire layouts are the caller's (tools/plonk_synth.py in the tests); the second
 * reading is that file's numpy generator.  The product's form is sipp_amd/csrc/witness.hip.
 */
#include "plonk.h"
#include "poseidon_constants.h"

static uint64_t pow7(uint64_t x) {
    const uint64_t x2 = gl_mul(x, x), x3 = gl_mul(x2, x), x4 = gl_mul(x2, x2);
    return gl_mul(x3, x4);
}

/* hash/poseidon.rs mds_layer: out[r] = sum_i s[(i + r) % 12] CIRC[i] + s[r] DIAG[r], every sum exact in 128 bits */
static void mds(uint64_t s[12]) {
    uint64_t o[12];
    for (int r = 0; r < 12; r++) {
        u128 acc = (u128)s[r] * POSEIDON_DIAG[r];
        for (int i = 0; i < 12; i++) acc += (u128)s[(i + r) % 12] * POSEIDON_CIRC[i];
        o[r] = gl_reduce128(acc);
    }
    for (int r = 0; r < 12; r++) s[r] = o[r];
}

static int layout_ok(const orc_plonk_generator *g, uint32_t nw, uint32_t nc) {
    const uint32_t *p = g->p;
    if (g->selector_index >= nc) return 0;
    switch (g->kind) {
    case 1: return 4ull * p[0] <= nw && p[1] < nc && p[2] < nc;
    case 2: return 1ull + p[0] <= nw && p[1] >= 1 && p[1] <= 32 && (uint64_t)p[0] * p[1] <= 64;
    case 3: return p[0] <= nw && (uint64_t)p[1] + p[0] <= nc;
    case 4: return nw >= 4;
    case 5: return p[2] <= 16 && p[1] >= 5 + 2 * p[2] && (uint64_t)p[0] * p[1] <= nw;
    case 6: return p[2] >= 1 && p[2] <= 6 && p[1] >= 2 + (1u << p[2]) + p[2] && (uint64_t)p[0] * p[1] <= nw;
    case 7: return 4ull + 3ull * p[0] <= nw;
    case 8: return (uint64_t)p[0] + 12 <= nw && (uint64_t)p[1] + 12 <= nw && (uint64_t)p[2] + 106 <= nw;
    default: return 0;
    }
}

#define W(j) wires[(size_t)(j) * n + i]
#define K(j) consts[(size_t)(j) * n + i]

static void run_row(uint64_t *wires, const uint64_t *consts, size_t n, size_t i, const orc_plonk_generator *g, const uint64_t pih[4]) {
    const uint32_t *p = g->p;
    switch (g->kind) {
    case 1:
        for (uint32_t k = 0; k < p[0]; k++)
            W(4 * k + 3) = gl_add(gl_mul(K(p[1]), gl_mul(W(4 * k), W(4 * k + 1))), gl_mul(K(p[2]), W(4 * k + 2)));
        break;
    case 2:
        for (uint32_t l = 0; l < p[0]; l++) W(1 + l) = (W(0) >> (p[1] * l)) & ((1ull << p[1]) - 1);
        break;
    case 3:
        for (uint32_t l = 0; l < p[0]; l++) W(l) = K(p[1] + l);
        break;
    case 4:
        for (uint32_t l = 0; l < 4; l++) W(l) = pih[l];
        break;
    case 5:
        for (uint32_t op = 0; op < p[0]; op++) {
            const uint32_t b = p[1] * op;
            const uint64_t full = (W(b) & 0xffffffffULL) * (W(b + 1) & 0xffffffffULL) + (W(b + 2) & 0xffffffffULL);
            W(b + 3) = full & 0xffffffffULL;
            W(b + 4) = full >> 32;
            for (uint32_t h = 0; h < 2; h++)
                for (uint32_t l = 0; l < p[2]; l++) W(b + 5 + p[2] * h + l) = (W(b + 3 + h) >> (2 * l)) & 3;
        }
        break;
    case 6:
        for (uint32_t cp = 0; cp < p[0]; cp++) {
            const uint32_t b = p[1] * cp, len = 1u << p[2];
            const uint32_t idx = (uint32_t)W(b) & (len - 1);
            W(b + 1) = W(b + 2 + idx);
            for (uint32_t l = 0; l < p[2]; l++) W(b + 2 + len + l) = (idx >> l) & 1;
        }
        break;
    case 7: {
        gl2 alpha = {W(0), W(1)}, acc = {W(2), W(3)};
        for (uint32_t l = 0; l < p[0]; l++) {
            /* (a0 + a1 X)(al0 + al1 X) mod X^2 - W, plus the base coefficient */
            gl2 nx;
            nx.c0 = gl_add(gl_add(gl_mul(acc.c0, alpha.c0), gl_mul(gl_mul(acc.c1, alpha.c1), p[1])), W(4 + l));
            nx.c1 = gl_add(gl_mul(acc.c0, alpha.c1), gl_mul(acc.c1, alpha.c0));
            W(4 + p[0] + 2 * l) = nx.c0;
            W(5 + p[0] + 2 * l) = nx.c1;
            acc = nx;
        }
        break;
    }
    case 8: {
        uint64_t s[12];
        for (int l = 0; l < 12; l++) s[l] = W(p[0] + l);
        for (uint32_t rnd = 0; rnd < 30; rnd++) {
            for (int l = 0; l < 12; l++) s[l] = gl_add(s[l], POSEIDON_RC[12 * rnd + l]);
            if (rnd < 4 || rnd >= 26) {
                const uint32_t base = rnd < 4 ? p[2] + 12 * (rnd - 1) : p[2] + 58 + 12 * (rnd - 26);
                for (int l = 0; l < 12; l++) {
                    if (rnd) W(base + l) = s[l];
                    s[l] = pow7(s[l]);
                }
            } else {
                W(p[2] + 36 + (rnd - 4)) = s[0];
                s[0] = pow7(s[0]);
            }
            mds(s);
        }
        for (int l = 0; l < 12; l++) W(p[1] + l) = s[l];
        break;
    }
    }
}

int orc_plonk_generate_witness(uint64_t *wires, const uint64_t *consts, unsigned log_n, uint32_t num_wires, uint32_t num_constants,
                               const orc_plonk_generator *gens, size_t n_gens, const uint64_t pih[4]) {
    if (!wires || !consts || (!gens && n_gens) || log_n < 1 || log_n > 26) return -1;
    const size_t n = (size_t)1 << log_n;
    for (size_t k = 0; k < n_gens; k++)
        if (!layout_ok(&gens[k], num_wires, num_constants) || (gens[k].kind == 4 && !pih)) return -1;
    for (size_t k = 0; k < n_gens; k++) {
        const orc_plonk_generator *g = &gens[k];
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n; i++)
            if (consts[(size_t)g->selector_index * n + i] == g->row) run_row(wires, consts, n, i, g, pih);
    }
    return 0;
}

int orc_plonk_generate_witness_levels(uint64_t *wires, const uint64_t *consts, unsigned log_n, uint32_t num_wires, uint32_t num_constants,
                                      const orc_plonk_generator *gens, size_t n_gens, const uint64_t pih[4], uint32_t n_levels,
                                      const uint32_t *rows, const uint32_t *level_offsets, const uint64_t *copy_src, const uint64_t *copy_dst,
                                      const uint32_t *copy_offsets) {
    if (!wires || !consts || (!gens && n_gens) || log_n < 1 || log_n > 26 || !n_levels || !rows || !level_offsets || !copy_offsets) return -1;
    const size_t n = (size_t)1 << log_n, cells = (size_t)num_wires * n;
    for (size_t k = 0; k < n_gens; k++)
        if (!layout_ok(&gens[k], num_wires, num_constants) || (gens[k].kind == 4 && !pih)) return -1;
    for (uint32_t l = 0; l < n_levels; l++)
        if (level_offsets[l] > level_offsets[l + 1] || copy_offsets[l] > copy_offsets[l + 1]) return -1;
    if (level_offsets[n_levels] > n || (copy_offsets[n_levels] && (!copy_src || !copy_dst))) return -1;
    for (uint32_t k = 0; k < level_offsets[n_levels]; k++)
        if (rows[k] >= n) return -1;
    for (uint32_t k = 0; k < copy_offsets[n_levels]; k++)
        if (copy_src[k] >= cells || copy_dst[k] >= cells) return -1;
    for (uint32_t l = 0; l < n_levels; l++) {
        for (uint32_t k = level_offsets[l]; k < level_offsets[l + 1]; k++) {
            const size_t i = rows[k];
            for (size_t g = 0; g < n_gens; g++)
                if (consts[(size_t)gens[g].selector_index * n + i] == gens[g].row) run_row(wires, consts, n, i, &gens[g], pih);
        }
        for (uint32_t k = copy_offsets[l]; k < copy_offsets[l + 1]; k++) wires[copy_dst[k]] = wires[copy_src[k]];
    }
    return 0;
}
