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

// one generator on one row
__device__ __forceinline__ void run_generator(uint64_t* wires, const uint64_t* consts, uint32_t n, uint32_t i, const sipp_plonk_generator& g,
                                              const uint64_t* pih) {
    auto W = [&](uint32_t j) -> uint64_t& { return wires[(size_t)j * n + i]; };
    auto K = [&](uint32_t j) -> uint64_t { return consts[(size_t)j * n + i]; };
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
        for (uint32_t l = 0; l < 4; l++) W(l) = pih[l];
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

__global__ void __launch_bounds__(256) plonk_witness_kernel(GenArgs a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, n = a.n;
    if (i >= n) return;
    if (a.consts[(size_t)a.g.selector_index * n + i] != a.g.row) return;
    run_generator(a.wires, a.consts, n, i, a.g, a.pih);
}

// ---- level by level -------------------------------------------------------------------------------------------------------------------
constexpr uint32_t MAX_GENS = 16;
constexpr uint32_t COOP_BELOW_ROWS = 16384;   // levels with fewer rows than this leave SIMDs empty with one lane per row
struct LevelArgs {
    uint64_t* wires;
    const uint64_t* consts;
    uint32_t n, n_gens;
    const uint32_t* rows;     // the level's rows
    uint32_t count;
    int* err;
    uint64_t pih[4];
    sipp_plonk_generator g[MAX_GENS];
};

// one lane per row of the level; the row runs the generator whose selector value it holds (the lanes of a wave may hold different gates:
// the families are short except Poseidon, whose rows a schedule keeps together)
__global__ void __launch_bounds__(64) plonk_witness_level_kernel(LevelArgs a) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.count) return;
    const uint32_t i = a.rows[k];
    if (i >= a.n) {
        *a.err = 1;
        return;
    }
    for (uint32_t q = 0; q < a.n_gens; q++)
        if (a.consts[(size_t)a.g[q].selector_index * a.n + i] == a.g[q].row) run_generator(a.wires, a.consts, a.n, i, a.g[q], a.pih);
}

// THIN levels (a hash chain's link: a few hundred to a few thousand rows) are latency-bound -- one lane walking a whole permutation is
// ~23 k dependent instructions.  Here a row gets SIXTEEN lanes (four rows per wave): lane l < 12 owns state element l of a Poseidon row,
// S-boxes side by side, the MDS layer through LDS (every lane reads the twelve elements of its row and accumulates its own output from
// exact 32-bit halves) -- ~3.5 k instructions deep; the short families run on lane 0 of their row's group.  pos = index of the Poseidon
// generator in a.g, or -1.  One wave per block: the two barriers per round cost nothing.
__global__ void __launch_bounds__(64) plonk_witness_level_coop_kernel(LevelArgs a, int pos) {
    __shared__ uint64_t sh[4][12];
    const uint32_t grp = threadIdx.x >> 4, l = threadIdx.x & 15, k = blockIdx.x * 4 + grp, n = a.n;
    bool ok = k < a.count;
    uint32_t i = ok ? a.rows[k] : 0;
    if (ok && i >= n) {
        if (l == 0) *a.err = 1;
        ok = false;
        i = 0;
    }
    bool is_pos = false;
    if (ok) {
        if (pos >= 0) is_pos = a.consts[(size_t)a.g[pos].selector_index * n + i] == a.g[pos].row;
        if (l == 0)
            for (uint32_t q = 0; q < a.n_gens; q++)
                if ((int)q != pos && a.consts[(size_t)a.g[q].selector_index * n + i] == a.g[q].row) run_generator(a.wires, a.consts, n, i, a.g[q], a.pih);
    }
    if (pos < 0 || !__syncthreads_or(is_pos)) return;        // block-uniform: no Poseidon row among the four
    const uint32_t in = a.g[pos].p[0], out = a.g[pos].p[1], sb = a.g[pos].p[2];
    const uint32_t e = l < 12 ? l : 0;                        // lanes 12 .. 15 shadow element 0 and never store
    const bool act = is_pos && l < 12;
    uint32_t coef[12];                                        // this lane's row of the MDS matrix
#pragma unroll
    for (int c = 0; c < 12; c++) coef[c] = w_mds_circ[(c + 12 - e) % 12] + ((e == 0 && c == 0) ? MDS_DIAG0 : 0);
    uint64_t s = act ? a.wires[(size_t)(in + e) * n + i] : 0;
#pragma unroll 1
    for (uint32_t rnd = 0; rnd < 30; rnd++) {
        const bool full = rnd < 4 || rnd >= 26;
        s = gl::add(s, w_rc[12 * rnd + e]);
        if (full || e == 0) {
            if (act && rnd) {
                const uint32_t w = full ? (rnd < 4 ? sb + 12 * (rnd - 1) : sb + 58 + 12 * (rnd - 26)) + e : sb + 36 + (rnd - 4);
                a.wires[(size_t)w * n + i] = s;
            }
            s = pow7(s);
        }
        if (l < 12) sh[grp][l] = s;
        __syncthreads();
        uint64_t al = 0, ah = 0;
#pragma unroll
        for (int c = 0; c < 12; c++) {
            const uint64_t v = sh[grp][c];
            al += (uint64_t)(uint32_t)v * coef[c];
            ah += (v >> 32) * coef[c];
        }
        const uint64_t lo = al + (ah << 32);
        s = gl::reduce96((uint32_t)(ah >> 32) + (lo < al ? 1u : 0u), lo);
        __syncthreads();
    }
    if (act) a.wires[(size_t)(out + e) * n + i] = s;
}

__global__ void __launch_bounds__(256) plonk_witness_copy_kernel(uint64_t* wires, const uint64_t* src, const uint64_t* dst, uint32_t count,
                                                                 uint64_t cells, int* err) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    const uint64_t s = src[k], d = dst[k];
    if (s >= cells || d >= cells) {
        *err = 1;
        return;
    }
    wires[d] = wires[s];
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

// argument checks shared by both entry points; uploads this translation unit's round constants when a Poseidon generator is there
static int witness_prepare(sipp_ctx* ctx, const uint64_t* d_wires, const uint64_t* d_constants, uint32_t log_n, uint32_t num_wires,
                           uint32_t num_constants, const sipp_plonk_generator* gens, size_t n_gens, const uint64_t* public_inputs_hash) {
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
    return SIPP_OK;
}

extern "C" int sipp_plonk_generate_witness(sipp_ctx* ctx, uint64_t* d_wires, const uint64_t* d_constants, uint32_t log_n, uint32_t num_wires,
                                           uint32_t num_constants, const sipp_plonk_generator* gens, size_t n_gens,
                                           const uint64_t public_inputs_hash[4]) {
    if (!ctx) return SIPP_E_BADARG;
    SIPP_TRY(witness_prepare(ctx, d_wires, d_constants, log_n, num_wires, num_constants, gens, n_gens, public_inputs_hash));
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

void sipp_witness_graph_release(sipp_ctx* ctx) {
    if (ctx->wgraph.exec) (void)hipGraphExecDestroy(ctx->wgraph.exec);
    if (ctx->wgraph.graph) (void)hipGraphDestroy(ctx->wgraph.graph);
    ctx->wgraph.exec = nullptr;
    ctx->wgraph.graph = nullptr;
    ctx->wgraph.key.clear();
}

extern "C" int sipp_plonk_generate_witness_levels(sipp_ctx* ctx, uint64_t* d_wires, const uint64_t* d_constants, uint32_t log_n,
                                                  uint32_t num_wires, uint32_t num_constants, const sipp_plonk_generator* gens, size_t n_gens,
                                                  const uint64_t public_inputs_hash[4], const sipp_plonk_schedule* sched) {
    if (!ctx) return SIPP_E_BADARG;
    SIPP_TRY(witness_prepare(ctx, d_wires, d_constants, log_n, num_wires, num_constants, gens, n_gens, public_inputs_hash));
    if (!sched || !sched->n_levels || sched->n_levels > (1u << 20) || !sched->d_rows || !sched->level_offsets || !sched->copy_offsets || n_gens > MAX_GENS)
        return sipp_fail(ctx, SIPP_E_BADARG, "plonk witness: no schedule, no level, or more than 16 generators");
    const uint32_t n = 1u << log_n, L = sched->n_levels;
    for (uint32_t l = 0; l < L; l++)
        if (sched->level_offsets[l] > sched->level_offsets[l + 1] || sched->copy_offsets[l] > sched->copy_offsets[l + 1])
            return sipp_fail(ctx, SIPP_E_BADARG, "plonk witness: schedule offsets must not decrease");
    if (sched->level_offsets[L] > n || (sched->copy_offsets[L] && (!sched->d_copy_src || !sched->d_copy_dst)))
        return sipp_fail(ctx, SIPP_E_BADARG, "plonk witness: more scheduled rows than the table has, or copies without their cell lists");
    // the error flag lives in a persistent one-word table of the ctx (the captured graph keeps its address)
    int* d_err = reinterpret_cast<int*>(sipp_table_get(ctx, 101, 0, 0));
    if (!d_err) {
        uint64_t* t = nullptr;
        SIPP_TRY(sipp_table_put(ctx, 101, 0, 0, std::vector<uint64_t>{0}, &t));
        d_err = reinterpret_cast<int*>(t);
    }
    int pos_gen = -1;
    for (size_t q = 0; q < n_gens; q++)
        if (gens[q].kind == SIPP_GEN_POSEIDON) pos_gen = (int)q;      // (two Poseidon layouts in one circuit: the last one gets the lanes)
    auto launch_all = [&]() -> hipError_t {
        (void)hipMemsetAsync(d_err, 0, sizeof(int), ctx->stream);
        for (uint32_t l = 0; l < L; l++) {
            const uint32_t r0 = sched->level_offsets[l], cnt = sched->level_offsets[l + 1] - r0;
            if (cnt) {
                LevelArgs a;
                a.wires = d_wires; a.consts = d_constants; a.n = n; a.n_gens = (uint32_t)n_gens; a.rows = sched->d_rows + r0; a.count = cnt; a.err = d_err;
                for (int q = 0; q < 4; q++) a.pih[q] = public_inputs_hash ? public_inputs_hash[q] : 0;
                for (size_t q = 0; q < n_gens; q++) a.g[q] = gens[q];
                if (cnt >= COOP_BELOW_ROWS)      // wide level: throughput, one lane per row
                    hipLaunchKernelGGL(plonk_witness_level_kernel, dim3((cnt + 63) / 64), dim3(64), 0, ctx->stream, a);
                else                             // thin level: latency, sixteen lanes per row
                    hipLaunchKernelGGL(plonk_witness_level_coop_kernel, dim3((cnt + 3) / 4), dim3(64), 0, ctx->stream, a, pos_gen);
            }
            const uint32_t c0 = sched->copy_offsets[l], cc = sched->copy_offsets[l + 1] - c0;
            if (cc)
                hipLaunchKernelGGL(plonk_witness_copy_kernel, dim3((cc + 255) / 256), dim3(256), 0, ctx->stream, d_wires, sched->d_copy_src + c0,
                                   sched->d_copy_dst + c0, cc, (uint64_t)num_wires << log_n, d_err);
        }
        return hipGetLastError();
    };
    const bool use_graph = !ctx->prof && !(ctx->kernel_routes & SIPP_ROUTE_WITNESS_NO_GRAPH);
    if (!use_graph) {
        ProfScope ps(ctx, "witness_levels");
        SIPP_CHECK_HIP(ctx, launch_all());
    } else {
        // key: everything the captured kernel arguments hold
        std::vector<uint64_t> key = {(uint64_t)(uintptr_t)d_wires, (uint64_t)(uintptr_t)d_constants, log_n, num_wires, num_constants, n_gens, L,
                                     (uint64_t)(uintptr_t)sched->d_rows, (uint64_t)(uintptr_t)sched->d_copy_src, (uint64_t)(uintptr_t)sched->d_copy_dst};
        for (int q = 0; q < 4; q++) key.push_back(public_inputs_hash ? public_inputs_hash[q] : 0);
        for (size_t q = 0; q < n_gens; q++) {
            const uint32_t* w = reinterpret_cast<const uint32_t*>(&gens[q]);
            for (size_t x = 0; x < sizeof(sipp_plonk_generator) / 4; x++) key.push_back(w[x]);
        }
        for (uint32_t l = 0; l <= L; l++) key.push_back(((uint64_t)sched->level_offsets[l] << 32) | sched->copy_offsets[l]);
        if (!ctx->wgraph.exec || ctx->wgraph.key != key) {
            sipp_witness_graph_release(ctx);
            SIPP_CHECK_HIP(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
            const hipError_t le = launch_all();
            hipGraph_t g = nullptr;
            const hipError_t ce = hipStreamEndCapture(ctx->stream, &g);
            if (le != hipSuccess || ce != hipSuccess || !g) {
                if (g) (void)hipGraphDestroy(g);
                return sipp_fail(ctx, SIPP_E_HIP, "plonk witness: capturing the level launches as a hipGraph failed");
            }
            ctx->wgraph.graph = g;
            if (hipGraphInstantiate(&ctx->wgraph.exec, g, nullptr, nullptr, 0) != hipSuccess) {
                sipp_witness_graph_release(ctx);
                return sipp_fail(ctx, SIPP_E_HIP, "plonk witness: hipGraphInstantiate failed");
            }
            ctx->wgraph.key = key;
        }
        SIPP_CHECK_HIP(ctx, hipGraphLaunch(ctx->wgraph.exec, ctx->stream));
    }
    int h_err = 0;
    SIPP_CHECK_HIP(ctx, hipMemcpyAsync(&h_err, d_err, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    SIPP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h_err) return sipp_fail(ctx, SIPP_E_BADARG, "plonk witness: the schedule names a row or a cell outside the wire table");
    return SIPP_OK;
}
