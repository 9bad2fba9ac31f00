// sipp_amd/csrc/witness.hip -- the gates' WITNESS GENERATORS of the outer plonky2 proof on the device:
// plonky2's prove() (reference src/verifier_circuit.rs:253 `data.prove(pw)`) begins with generate_partial_witness (iop/generator.rs), where
// every gate instance's SimpleGenerator computes the wires its constraints determine.  plonky2 @ InternetMaximalism/plonky2 541e127 is not
// vendored: the gate families and their generators follow the recalled upstream gates (gates/arithmetic_base.rs, base_sum.rs, constant.rs,
// public_input.rs, random_access.rs, reducing.rs, poseidon.rs; plonky2_u32 gates/arithmetic_u32.rs) in the layouts the caller passes as
// data (include/sipp_hip.h, sipp_plonk_generator); the checker is oracle/plonk_gates.c (orc_plonk_generate_witness) and the numpy
// generator of tools/plonk_synth.py.
//
// Layout: the wire table [num_wires][N] the prover reads next, natural row order; one LANE PER ROW, one launch per generator; a lane whose
// selector cell does not hold the generator's gate index leaves at once.  Consecutive lanes = consecutive rows of one column: every load
// and store is coalesced; nothing is staged.  The Poseidon generator is the only one with real arithmetic (the naive 30-round form: the
// S-box INPUTS of every round are wires, so the lazy partial-round form of the hash kernels does not apply): 12 x 12 small-constant
// products per round on exactly accumulated 32-bit halves, one reduction per output.
#include "ctx.hpp"
#include "poseidon_constants.h"
#include <mutex>

namespace {

__constant__ uint64_t w_rc[360];
__constant__ uint32_t w_mds_circ[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
constexpr uint32_t MDS_DIAG0 = 8;

struct GenArgs {
    uint64_t* wires;
    const uint64_t* consts;
    uint32_t n;
    sipp_plonk_generator g;
    uint64_t pih[4];
};

__device__ __forceinline__ uint64_t pow7(uint64_t x) {
    const uint64_t x2 = gl::mul(x, x), x4 = gl::mul(x2, x2);
    return gl::mul(gl::mul(x4, x2), x);
}

// out = M s: M[r][c] = circ[(c - r) mod 12] + [r = c = 0] 8, entries below 2^6: the halves accumulate exactly (12 products below 2^38)
__device__ __forceinline__ void mds(uint64_t (&s)[12]) {
    uint64_t out[12];
#pragma unroll
    for (int r = 0; r < 12; r++) {
        uint64_t al = 0, ah = 0;
#pragma unroll
        for (int c = 0; c < 12; c++) {
            const uint32_t k = w_mds_circ[(c - r + 12) % 12] + ((r == 0 && c == 0) ? MDS_DIAG0 : 0);
            al += (uint64_t)(uint32_t)s[c] * k;
            ah += (s[c] >> 32) * k;
        }
        // al + 2^32 ah as (hi32, lo): ah < 2^42
        const uint64_t lo = al + (ah << 32);
        const uint32_t hi = (uint32_t)(ah >> 32) + (lo < al ? 1u : 0u);
        out[r] = gl::reduce96(hi, lo);
    }
#pragma unroll
    for (int r = 0; r < 12; r++) s[r] = out[r];
}

__global__ void __launch_bounds__(256) plonk_witness_kernel(GenArgs a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, n = a.n;
    if (i >= n) return;
    const sipp_plonk_generator g = a.g;
    if (a.consts[(size_t)g.selector_index * n + i] != g.row) return;
    auto W = [&](uint32_t j) -> uint64_t& { return a.wires[(size_t)j * n + i]; };
    auto K = [&](uint32_t j) -> uint64_t { return a.consts[(size_t)j * n + i]; };
    switch (g.kind) {
    case SIPP_GEN_ARITHMETIC: {
        const uint64_t c0 = K(g.p[1]), c1 = K(g.p[2]);
        for (uint32_t k = 0; k < g.p[0]; k++)
            W(4 * k + 3) = gl::add(gl::mul(c0, gl::mul(W(4 * k), W(4 * k + 1))), gl::mul(c1, W(4 * k + 2)));
        break;
    }
    case SIPP_GEN_BASE_SPLIT: {
        const uint64_t v = W(0), mask = (1ull << g.p[1]) - 1;
        for (uint32_t l = 0; l < g.p[0]; l++) W(1 + l) = (v >> (g.p[1] * l)) & mask;
        break;
    }
    case SIPP_GEN_CONSTANT:
        for (uint32_t l = 0; l < g.p[0]; l++) W(l) = K(g.p[1] + l);
        break;
    case SIPP_GEN_PUBLIC_INPUT:
        for (uint32_t l = 0; l < 4; l++) W(l) = a.pih[l];
        break;
    case SIPP_GEN_U32_MUL_ADD:
        for (uint32_t op = 0; op < g.p[0]; op++) {
            const uint32_t b = g.p[1] * op, L = g.p[2];
            // the inputs are u32 values (range-checked where they were produced): the full result fits 64 bits
            const uint64_t full = (W(b) & 0xffffffffull) * (W(b + 1) & 0xffffffffull) + (W(b + 2) & 0xffffffffull);
            const uint64_t half[2] = {full & 0xffffffffull, full >> 32};
            W(b + 3) = half[0];
            W(b + 4) = half[1];
            for (uint32_t h = 0; h < 2; h++)
                for (uint32_t l = 0; l < L; l++) W(b + 5 + L * h + l) = (half[h] >> (2 * l)) & 3;
        }
        break;
    case SIPP_GEN_RANDOM_ACCESS:
        for (uint32_t cp = 0; cp < g.p[0]; cp++) {
            const uint32_t b = g.p[1] * cp, bits = g.p[2], len = 1u << bits;
            const uint32_t idx = (uint32_t)W(b) & (len - 1);
            W(b + 1) = W(b + 2 + idx);
            for (uint32_t l = 0; l < bits; l++) W(b + 2 + len + l) = (idx >> l) & 1;
        }
        break;
    case SIPP_GEN_REDUCING: {
        const uint32_t Kc = g.p[0];
        const uint64_t nr = g.p[1], al0 = W(0), al1 = W(1);
        uint64_t a0 = W(2), a1 = W(3);
        for (uint32_t l = 0; l < Kc; l++) {
            const uint64_t n0 = gl::add(gl::add(gl::mul(a0, al0), gl::mul(gl::mul(a1, al1), nr)), W(4 + l));
            const uint64_t n1 = gl::add(gl::mul(a0, al1), gl::mul(a1, al0));
            W(4 + Kc + 2 * l) = n0;
            W(5 + Kc + 2 * l) = n1;
            a0 = n0, a1 = n1;
        }
        break;
    }
    case SIPP_GEN_POSEIDON: {
        const uint32_t in = g.p[0], out = g.p[1], sb = g.p[2];
        uint64_t s[12];
#pragma unroll
        for (int l = 0; l < 12; l++) s[l] = W(in + l);
#pragma unroll 1
        for (uint32_t rnd = 0; rnd < 30; rnd++) {
            const bool full = rnd < 4 || rnd >= 26;
#pragma unroll
            for (int l = 0; l < 12; l++) s[l] = gl::add(s[l], w_rc[12 * rnd + l]);
            if (full) {
                const uint32_t base = rnd < 4 ? sb + 12 * (rnd - 1) : sb + 58 + 12 * (rnd - 26);
#pragma unroll
                for (int l = 0; l < 12; l++) {
                    if (rnd) W(base + l) = s[l];
                    s[l] = pow7(s[l]);
                }
            } else {
                W(sb + 36 + (rnd - 4)) = s[0];
                s[0] = pow7(s[0]);
            }
            mds(s);
        }
#pragma unroll
        for (int l = 0; l < 12; l++) W(out + l) = s[l];
        break;
    }
    default: break;
    }
}

// the wires a generator reads or writes stay inside the table; constant columns inside the constants
bool layout_ok(const sipp_plonk_generator& g, uint32_t num_wires, uint32_t num_constants) {
    if (g.selector_index >= num_constants) return false;
    const uint64_t nw = num_wires;
    switch (g.kind) {
    case SIPP_GEN_ARITHMETIC: return 4ull * g.p[0] <= nw && g.p[1] < num_constants && g.p[2] < num_constants;
    case SIPP_GEN_BASE_SPLIT: return 1ull + g.p[0] <= nw && g.p[1] >= 1 && g.p[1] <= 32 && (uint64_t)g.p[0] * g.p[1] <= 64;
    case SIPP_GEN_CONSTANT: return g.p[0] <= nw && (uint64_t)g.p[1] + g.p[0] <= num_constants;
    case SIPP_GEN_PUBLIC_INPUT: return 4 <= nw;
    case SIPP_GEN_U32_MUL_ADD: return g.p[2] <= 16 && g.p[1] >= 5 + 2 * g.p[2] && (uint64_t)g.p[0] * g.p[1] <= nw;
    case SIPP_GEN_RANDOM_ACCESS: return g.p[2] >= 1 && g.p[2] <= 6 && g.p[1] >= 2 + (1u << g.p[2]) + g.p[2] && (uint64_t)g.p[0] * g.p[1] <= nw;
    case SIPP_GEN_REDUCING: return 4ull + 3ull * g.p[0] <= nw;
    case SIPP_GEN_POSEIDON: return (uint64_t)g.p[0] + 12 <= nw && (uint64_t)g.p[1] + 12 <= nw && (uint64_t)g.p[2] + 106 <= nw;
    default: return false;
    }
}

const char* gen_name(uint32_t kind) {
    static const char* names[] = {"", "witness_arithmetic", "witness_base_split", "witness_constant", "witness_public_input", "witness_u32",
                                  "witness_random_access", "witness_reducing", "witness_poseidon"};
    return kind <= SIPP_GEN_POSEIDON ? names[kind] : "witness";
}

}  // namespace

extern "C" int sipp_plonk_generate_witness(sipp_ctx* ctx, uint64_t* d_wires, const uint64_t* d_constants, uint32_t log_n, uint32_t num_wires,
                                           uint32_t num_constants, const sipp_plonk_generator* gens, size_t n_gens,
                                           const uint64_t public_inputs_hash[4]) {
    if (!ctx) return SIPP_E_BADARG;
    if (!d_wires || !d_constants || (!gens && n_gens) || log_n < 1 || log_n > 26 || !num_wires || !num_constants)
        return sipp_fail(ctx, SIPP_E_BADARG, "plonk witness: null table, no columns or log_n outside 1 .. 26");
    bool needs_pih = false, needs_rc = false;
    for (size_t k = 0; k < n_gens; k++) {
        if (!layout_ok(gens[k], num_wires, num_constants))
            return sipp_fail(ctx, SIPP_E_BADARG, "plonk witness: a generator's layout leaves the wire table / the constants, or its family is unknown");
        needs_pih |= gens[k].kind == SIPP_GEN_PUBLIC_INPUT;
        needs_rc |= gens[k].kind == SIPP_GEN_POSEIDON;
    }
    if (needs_pih && !public_inputs_hash) return sipp_fail(ctx, SIPP_E_BADARG, "plonk witness: a PublicInput generator without the public-inputs hash");
    SIPP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    if (needs_rc) {      // the round constants of this translation unit, once per device (a blocking copy: complete before any launch below)
        static std::mutex mu;
        static bool done[64] = {false};
        std::lock_guard<std::mutex> lk(mu);
        const int dev = ctx->device & 63;
        if (!done[dev]) {
            SIPP_CHECK_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(w_rc), SIPP_POSEIDON_RC, sizeof(SIPP_POSEIDON_RC)));
            done[dev] = true;
        }
    }
    const uint32_t n = 1u << log_n;
    for (size_t k = 0; k < n_gens; k++) {
        GenArgs a;
        a.wires = d_wires; a.consts = d_constants; a.n = n; a.g = gens[k];
        for (int l = 0; l < 4; l++) a.pih[l] = public_inputs_hash ? public_inputs_hash[l] : 0;
        ProfScope ps(ctx, gen_name(gens[k].kind));
        hipLaunchKernelGGL(plonk_witness_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, a);
        SIPP_CHECK_HIP(ctx, hipGetLastError());
    }
    return SIPP_OK;
}
