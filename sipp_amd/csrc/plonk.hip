// sipp_amd/csrc/plonk.hip -- plonky2's WIRE PERMUTATION ARGUMENT on the device (the protocol-generic part of
// `data.prove(pw)` at reference src/verifier_circuit.rs:253 that needs no circuit): Z and the partial products
// (plonk/prover.rs wires_permutation_partial_products_and_zs), the permutation terms of the vanishing polynomial reduced with the
// powers of every alpha and divided by Z_H on the quotient coset (plonk/vanishing_poly.rs eval_vanishing_poly_base_batch,
// compute_quotient_polys), for any number of routed wires, chunk size (= quotient degree factor, a power of two) and challenges.
// plonky2 @ InternetMaximalism/plonky2 541e127 is not vendored: the sequence follows oracle/plonk.c (and its second reading in
// oracle/py/plonky2_generic.py); the GATE constraints of the reference's circuit are not part of this.
//
// Layout: everything column-major [column][row] like the rest of the library; LDEs in leaf order, so the quotient coset
// 7 <w_(N D)> is simply the FIRST N D leaves of a blowup-2^rate_bits LDE (leaf t = coset index bitrev(t)), and the quotient values are
// written in leaf order, which is what the coset iNTT (bit-reversed in, natural out) reads.
// Kernels: one lane per row (chunk quotients with ONE inversion per row and challenge: Montgomery's trick over the <= 32 chunk
// denominators), a two-level prefix product over the rows for Z, one lane per coset point for the quotient.  All HBM-streaming with
// O(R) products per cell read -- integer VALU bound like the STARK quotient kernels.
#include "ctx.hpp"
#include "prover.hpp"
#include <algorithm>
#include <map>

namespace {

constexpr uint32_t MAX_CHUNKS = 32, MAX_CH = 8;

struct ZsArgs {
    const uint64_t* wires;    // [R][n]
    const uint64_t* sigmas;   // [R][n]
    uint32_t log_n, R, D, m, C;
    const uint64_t* bk;       // [C][R]  beta_c 7^j
    uint64_t beta[MAX_CH], gamma[MAX_CH];
    uint64_t w;               // primitive n-th root
    uint64_t* chunk;          // [C][m][n]   quotient_chunk_products
    uint64_t* tot;            // [C][n]      product of a row's chunks = Z(g x) / Z(x)
};

__global__ void __launch_bounds__(256) plonk_chunk_kernel(ZsArgs a) {
    const uint32_t n = 1u << a.log_n, i = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
    if (i >= n) return;
    const uint64_t x = gl::pow(a.w, i), beta = a.beta[c], gamma = a.gamma[c];
    const uint64_t* bk = a.bk + (size_t)c * a.R;
    uint64_t num[MAX_CHUNKS], pre[MAX_CHUNKS];   // chunk numerators; prefix products of the chunk denominators
    uint64_t run = 1;
#pragma unroll 1
    for (uint32_t q = 0; q < a.m; q++) {
        uint64_t nu = 1, de = 1;
        const uint32_t j1 = min((q + 1) * a.D, a.R);
        for (uint32_t j = q * a.D; j < j1; j++) {
            const uint64_t wv = a.wires[(size_t)j * n + i];
            nu = gl::mul(nu, gl::add(gl::add(wv, gl::mul(bk[j], x)), gamma));
            de = gl::mul(de, gl::add(gl::add(wv, gl::mul(beta, a.sigmas[(size_t)j * n + i])), gamma));
        }
        num[q] = nu;
        pre[q] = run;                 // product of the denominators before chunk q
        run = gl::mul(run, de);
        // keep the denominator itself in `chunk` for the backward sweep
        a.chunk[((size_t)c * a.m + q) * n + i] = de;
    }
    uint64_t inv = gl::inv(run), t = 1;
#pragma unroll 1
    for (uint32_t q = a.m; q-- > 0;) {
        uint64_t* cell = a.chunk + ((size_t)c * a.m + q) * n + i;
        const uint64_t de = *cell;
        const uint64_t qv = gl::mul(num[q], gl::mul(inv, pre[q]));   // num_q / den_q
        inv = gl::mul(inv, de);
        *cell = qv;
        t = gl::mul(t, qv);
    }
    a.tot[(size_t)c * n + i] = t;
}

// exclusive prefix product over the rows: one block per challenge, every thread owns n / 1024 consecutive rows
__global__ void __launch_bounds__(1024) plonk_scan_kernel(const uint64_t* tot, uint64_t* zs, uint32_t log_n) {
    __shared__ uint64_t part[1024];
    const uint32_t n = 1u << log_n, c = blockIdx.x, T = blockDim.x;
    const uint32_t per = (n + T - 1) / T, lo = threadIdx.x * per, hi = min(lo + per, n);
    const uint64_t* t = tot + (size_t)c * n;
    uint64_t* z = zs + (size_t)c * n;
    uint64_t acc = 1;
    for (uint32_t i = lo; i < hi; i++) acc = gl::mul(acc, t[i]);
    part[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t d = 1; d < T; d <<= 1) {          // Hillis-Steele inclusive scan of the 1024 partial products
        const uint64_t v = threadIdx.x >= d ? part[threadIdx.x - d] : 1;
        __syncthreads();
        part[threadIdx.x] = gl::mul(part[threadIdx.x], v);
        __syncthreads();
    }
    acc = threadIdx.x ? part[threadIdx.x - 1] : 1;
    for (uint32_t i = lo; i < hi; i++) {
        z[i] = acc;                                   // Z(w^i) = product of the rows before i; Z(1) = 1
        acc = gl::mul(acc, t[i]);
    }
}

__global__ void __launch_bounds__(256) plonk_pp_kernel(const uint64_t* chunk, uint64_t* out, uint32_t log_n, uint32_t m, uint32_t C) {
    const uint32_t n = 1u << log_n, i = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, np = m - 1;
    if (i >= n) return;
    uint64_t acc = out[(size_t)c * n + i];
    for (uint32_t q = 0; q < np; q++) {
        acc = gl::mul(acc, chunk[((size_t)c * m + q) * n + i]);
        out[((size_t)C + (size_t)c * np + q) * n + i] = acc;
    }
}

struct QuotArgs {
    const uint64_t* wl;       // [R][stride]  leaf order
    const uint64_t* sl;
    const uint64_t* zl;       // [C (1 + np)][stride]
    size_t stride;            // n << rate_bits
    uint32_t log_n, rate_bits, log_d, R, D, m, C;
    const uint64_t* bk;       // [C][R]
    uint64_t beta[MAX_CH], gamma[MAX_CH], alpha[MAX_CH];
    uint64_t w_nd;            // primitive (n D)-th root
    uint64_t zh_inv[64];      // 1 / (x^n - 1) by coset index mod D
    uint64_t zh[64];
    uint64_t n_field;         // n as a field element
    uint64_t* qv;             // [C][n D]  leaf order of the quotient coset
    const uint64_t* gt;       // [n_gt][stride] caller-supplied gate-constraint terms, leaf order (nullptr: none)
    uint32_t n_gt;
    // gates as data (sipp_plonk_prove_gates): the circuit's gate set interpreted at every point of the quotient coset
    const uint64_t* cl;       // [num_constants][stride] constant columns (selectors first), leaf order; nullptr: no gate set
    const sipp_plonk_gate* gates;
    // the gate programs COMPILED for this proof (compile_gates below): per gate its DISTINCT monomials, each evaluated once per point and
    // multiplied by ONE coefficient per challenge -- the alpha-folded sum of every place the monomial occurs in the gate's constraints
    const int64_t* cprog;     // per gate: n_distinct, then per monomial: n_factors, (kind, index) x n_factors
    const uint32_t* coff;     // [n_gates] word offset of a gate in cprog
    const uint32_t* boff;     // [n_gates] index of a gate's first monomial in cB
    const uint64_t* cB;       // [C][n_dist]
    uint32_t n_gates, many_sel, n_dist;
    uint64_t pih[4];
};

__global__ void __launch_bounds__(256) plonk_quotient_kernel(QuotArgs a) {
    const uint32_t lq = a.log_n + a.log_d, nd = 1u << lq, pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= nd) return;
    const uint32_t L = a.log_n + a.rate_bits;
    const uint32_t i = gl::bitrev(pos, lq);                              // coset index: x = 7 w^i
    const uint32_t nat = gl::bitrev(pos, L);                             // natural index in the blowup-2^rate_bits LDE
    const uint32_t pos_next = gl::bitrev((nat + (1u << a.rate_bits)) & ((1u << L) - 1), L);   // g x: D steps on the coset
    const uint64_t x = gl::mul(gl::GEN, gl::pow(a.w_nd, i));
    const uint64_t zh = a.zh[i & (a.D - 1)], zhi = a.zh_inv[i & (a.D - 1)];
    const uint64_t l0 = gl::mul(zh, gl::inv(gl::mul(a.n_field, gl::sub(x, 1))));
    const uint32_t np = a.m - 1, C = a.C;
    uint64_t acc[MAX_CH], apow[MAX_CH];
#pragma unroll
    for (uint32_t c = 0; c < MAX_CH; c++) acc[c] = 0, apow[c] = 1;
    auto push = [&](uint64_t term) {                                     // reduce_with_powers: sum_k alpha^k term_k, for every alpha
#pragma unroll
        for (uint32_t c = 0; c < MAX_CH; c++)
            if (c < C) {
                acc[c] = gl::add(acc[c], gl::mul(apow[c], term));
                apow[c] = gl::mul(apow[c], a.alpha[c]);
            }
    };
    for (uint32_t c = 0; c < C; c++) push(gl::mul(l0, gl::sub(a.zl[(size_t)c * a.stride + pos], 1)));   // L_0(x) (Z(x) - 1)
    for (uint32_t c = 0; c < C; c++) {
        const uint64_t* bk = a.bk + (size_t)c * a.R;
        uint64_t prev = a.zl[(size_t)c * a.stride + pos];
        for (uint32_t q = 0; q < a.m; q++) {
            uint64_t nu = 1, de = 1;
            const uint32_t j1 = min((q + 1) * a.D, a.R);
            for (uint32_t j = q * a.D; j < j1; j++) {
                const uint64_t wv = a.wl[(size_t)j * a.stride + pos];
                nu = gl::mul(nu, gl::add(gl::add(wv, gl::mul(bk[j], x)), a.gamma[c]));
                de = gl::mul(de, gl::add(gl::add(wv, gl::mul(a.beta[c], a.sl[(size_t)j * a.stride + pos])), a.gamma[c]));
            }
            const uint64_t next = q == np ? a.zl[(size_t)c * a.stride + pos_next] : a.zl[((size_t)C + (size_t)c * np + q) * a.stride + pos];
            push(gl::sub(gl::mul(prev, nu), gl::mul(next, de)));         // check_partial_products
            prev = next;
        }
    }
    // constraint_terms: the circuit's gate constraints at this point, evaluated by the caller (vanishing_poly.rs appends them to the
    // permutation terms; one reduce_with_powers runs over all of them)
    for (uint32_t k = 0; k < a.n_gt; k++) push(gl::canon(a.gt[(size_t)k * a.stride + pos]));
    if (a.cl) {
        // evaluate_gate_constraints with the gate set as data: term_j = sum_g filter_g c_{g,j}, so per challenge
        //   sum_j alpha^(n0 + j) term_j = sum_g filter_g (sum_j alpha^(n0 + j) c_{g,j});  the walk over gates, constraints, monomials and
        // factors is wave-uniform (every lane runs the same program on its own point), operands are committed LDE cells (canonical)
        auto operand = [&](int64_t kind, int64_t idx) -> uint64_t {
            return kind == 0 ? a.wl[(size_t)idx * a.stride + pos] : kind == 1 ? a.cl[(size_t)idx * a.stride + pos] : a.pih[idx];
        };
        for (uint32_t g = 0; g < a.n_gates; g++) {
            const sipp_plonk_gate ga = a.gates[g];
            if (!ga.num_constraints) continue;
            const uint64_t sv = a.cl[(size_t)ga.selector_index * a.stride + pos];
            uint64_t f = 1;                                              // compute_filter (gates/selectors.rs)
            for (uint32_t i = ga.group_lo; i < ga.group_hi; i++)
                if (i != ga.row) f = gl::mul(f, gl::sub((uint64_t)i, sv));
            if (a.many_sel) f = gl::mul(f, gl::sub(0xffffffffull, sv));
            uint64_t part[MAX_CH];
#pragma unroll
            for (uint32_t c = 0; c < MAX_CH; c++) part[c] = 0;
            // sum_j alpha^(n0 + j) c_{g,j} = sum over the gate's DISTINCT monomials d of  value_d(point) * B_d,  B_d = sum over the places
            // (constraint j, coefficient) where d occurs of alpha^(n0 + j) * coefficient (host, per proof): the twelve constraints of a
            // Poseidon round share their twelve S-box terms -- 2999 monomial instances of that gate are 309 distinct ones
            const int64_t* w = a.cprog + a.coff[g];
            const uint64_t* B = a.cB + a.boff[g];
            const uint32_t nd_g = (uint32_t)*w++;
            int64_t pw_kind = -1, pw_idx = -1;                            // the last pure power evaluated: operand, exponent, value
            int pw_exp = 0;
            uint64_t pw_val = 0;
            for (uint32_t d = 0; d < nd_g; d++) {
                const int nf = (int)*w++;
                uint64_t t;
                if (nf == 0) {
                    t = 1;
                } else {
                    // a PURE POWER of one operand (w^7 of a Poseidon gate; w, w^2 .. w^7 of (in + rc)^7, adjacent after the compile step's
                    // sort): one load, the power by squaring -- or ONE product when the previous monomial was the next lower power of the
                    // same operand.  Wave-uniform decisions (program words).
                    const int64_t k0 = w[0], i0 = w[1];
                    bool pure = true;
                    for (int q = 1; q < nf && pure; q++) pure = w[2 * q] == k0 && w[2 * q + 1] == i0;
                    if (pure) {
                        if (k0 == pw_kind && i0 == pw_idx && nf == pw_exp + 1) {
                            t = gl::mul(pw_val, operand(k0, i0));
                        } else {
                            const uint64_t x = operand(k0, i0);
                            t = x;
                            for (int bit = 30 - __builtin_clz((unsigned)nf); bit >= 0; bit--) {
                                t = gl::mul(t, t);
                                if ((nf >> bit) & 1) t = gl::mul(t, x);
                            }
                        }
                        pw_kind = k0; pw_idx = i0; pw_exp = nf; pw_val = t;
                        w += 2 * nf;
                    } else {
                        t = operand(k0, i0);
                        w += 2;
                        for (int q = 1; q < nf; q++, w += 2) t = gl::mul(t, operand(w[0], w[1]));
                    }
                }
#pragma unroll
                for (uint32_t c = 0; c < MAX_CH; c++)
                    if (c < C) part[c] = gl::mad(B[(size_t)c * a.n_dist + d], t, part[c]);
            }
#pragma unroll
            for (uint32_t c = 0; c < MAX_CH; c++)
                if (c < C) acc[c] = gl::mad(f, part[c], acc[c]);
        }
    }
    for (uint32_t c = 0; c < C; c++) a.qv[(size_t)c * nd + pos] = gl::mul(acc[c], zhi);
}

int check(sipp_ctx* ctx, const sipp_plonk_params* p, uint32_t log_n, uint32_t* log_d, uint32_t* m) {
    if (!p || p->num_routed_wires == 0 || p->num_challenges == 0 || log_n < 1 || log_n > 24) return sipp_fail(ctx, SIPP_E_BADARG, "plonk: bad parameters");
    if (p->max_degree < 2 || p->max_degree > 64)      // before the loop below: 1u << 32 is undefined
        return sipp_fail(ctx, SIPP_E_UNSUPPORTED, "plonk: the chunk size (quotient degree factor) must be a power of two in 2 .. 64");
    uint32_t ld = 0;
    while ((1u << ld) < p->max_degree) ld++;
    if ((1u << ld) != p->max_degree)
        return sipp_fail(ctx, SIPP_E_UNSUPPORTED, "plonk: the chunk size (quotient degree factor) must be a power of two in 2 .. 64");
    const uint32_t chunks = (p->num_routed_wires + p->max_degree - 1) / p->max_degree;
    if (chunks > MAX_CHUNKS || p->num_challenges > MAX_CH)
        return sipp_fail(ctx, SIPP_E_UNSUPPORTED, "plonk: at most 32 chunks of routed wires and 8 challenges");
    *log_d = ld;
    *m = chunks;
    return SIPP_OK;
}

// beta_c 7^j on the device (released with the caller's arena scope)
uint64_t* upload_bk(sipp_ctx* ctx, const sipp_plonk_params* p, const uint64_t* betas) {
    const uint32_t R = p->num_routed_wires, C = p->num_challenges;
    std::vector<uint64_t> h((size_t)C * R);
    uint64_t k = 1;
    for (uint32_t j = 0; j < R; j++) {
        for (uint32_t c = 0; c < C; c++) h[(size_t)c * R + j] = gl::mul(gl::canon(betas[c]), k);
        k = gl::mul(k, gl::GEN);
    }
    uint64_t* d = arena_alloc_t<uint64_t>(ctx, h.size());
    if (!d) return nullptr;
    if (hipMemcpyAsync(d, h.data(), h.size() * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess)   // `h` goes out of scope
        return nullptr;
    return d;
}

}  // namespace

uint32_t sipp_plonk_num_partial_products(const sipp_plonk_params* p) {
    if (!p || p->max_degree == 0 || p->num_routed_wires == 0) return 0;
    return (p->num_routed_wires + p->max_degree - 1) / p->max_degree - 1;
}

int sipp_plonk_zs_partial_products(sipp_ctx* ctx, const uint64_t* d_wires, const uint64_t* d_sigmas, uint32_t log_n, const sipp_plonk_params* p,
                                   const uint64_t* betas, const uint64_t* gammas, uint64_t* d_out) {
    if (!ctx || !d_wires || !d_sigmas || !betas || !gammas || !d_out) return SIPP_E_BADARG;
    uint32_t log_d, m;
    SIPP_TRY(check(ctx, p, log_n, &log_d, &m));
    SIPP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    ArenaScope scope(ctx);
    const size_t n = (size_t)1 << log_n;
    const uint32_t C = p->num_challenges;
    ZsArgs a{};
    a.wires = d_wires; a.sigmas = d_sigmas; a.log_n = log_n; a.R = p->num_routed_wires; a.D = p->max_degree; a.m = m; a.C = C;
    a.w = gl::root_of_unity(log_n);
    for (uint32_t c = 0; c < C; c++) { a.beta[c] = gl::canon(betas[c]); a.gamma[c] = gl::canon(gammas[c]); }
    a.bk = upload_bk(ctx, p, betas);
    a.chunk = arena_alloc_t<uint64_t>(ctx, (size_t)C * m * n);
    a.tot = arena_alloc_t<uint64_t>(ctx, (size_t)C * n);
    if (!a.bk || !a.chunk || !a.tot) return SIPP_E_NOMEM;
    {
        ProfScope ps(ctx, "plonk_zs");
        hipLaunchKernelGGL(plonk_chunk_kernel, dim3((unsigned)((n + 255) / 256), C), dim3(256), 0, ctx->stream, a);
        hipLaunchKernelGGL(plonk_scan_kernel, dim3(C), dim3(1024), 0, ctx->stream, a.tot, d_out, log_n);
        hipLaunchKernelGGL(plonk_pp_kernel, dim3((unsigned)((n + 255) / 256), C), dim3(256), 0, ctx->stream, a.chunk, d_out, log_n, m, C);
    }
    SIPP_CHECK_HIP(ctx, hipGetLastError());
    return sipp_sync(ctx);   // the scratch goes back with the scope
}

int sipp_plonk_quotient_chunks(sipp_ctx* ctx, const uint64_t* d_wires_lde, const uint64_t* d_sigmas_lde, const uint64_t* d_zs_lde, uint32_t log_n,
                               uint32_t rate_bits, const sipp_plonk_params* p, const uint64_t* betas, const uint64_t* gammas,
                               const uint64_t* alphas, uint64_t* d_chunks) {
    return sipp_plonk_quotient_chunks_ex(ctx, d_wires_lde, d_sigmas_lde, d_zs_lde, log_n, rate_bits, p, betas, gammas, alphas, nullptr, 0, d_chunks);
}

namespace {
// the gate set on the device: the constant columns' LDE, the circuit (host description, validated by the caller) and the hash of the public inputs
struct GateSet {
    const uint64_t* d_consts_lde;
    const sipp_plonk_circuit* c;
    const uint64_t* pih;
};
int quotient_chunks_impl(sipp_ctx* ctx, const uint64_t* d_wires_lde, const uint64_t* d_sigmas_lde, const uint64_t* d_zs_lde, uint32_t log_n,
                         uint32_t rate_bits, const sipp_plonk_params* p, const uint64_t* betas, const uint64_t* gammas, const uint64_t* alphas,
                         const uint64_t* d_gate_terms, uint32_t num_gate_terms, const GateSet* gs, uint64_t* d_chunks);
}  // namespace

int sipp_plonk_quotient_chunks_ex(sipp_ctx* ctx, const uint64_t* d_wires_lde, const uint64_t* d_sigmas_lde, const uint64_t* d_zs_lde, uint32_t log_n,
                                  uint32_t rate_bits, const sipp_plonk_params* p, const uint64_t* betas, const uint64_t* gammas,
                                  const uint64_t* alphas, const uint64_t* d_gate_terms, uint32_t num_gate_terms, uint64_t* d_chunks) {
    return quotient_chunks_impl(ctx, d_wires_lde, d_sigmas_lde, d_zs_lde, log_n, rate_bits, p, betas, gammas, alphas, d_gate_terms, num_gate_terms,
                                nullptr, d_chunks);
}

namespace {
// The gate programs compiled for one proof.  A gate's constraints are sums of monomials over the row's wires / constants / public-inputs
// hash, and the prover needs only  sum_j alpha^(n0 + j) constraint_j  per challenge: exchanging the sums, every DISTINCT monomial of a gate
// is evaluated once per point and weighted by B = sum over its occurrences (constraint j, coefficient) of alpha^(n0 + j) coefficient.
// Distinct = equal factor multisets.  Order inside a gate: pure powers first, by operand and ascending exponent (the kernel turns a run
// w, w^2, .. into one product each), then the mixed monomials.  (circuit_check has validated the program words.)
struct CompiledGates {
    std::vector<int64_t> prog;     // per gate: n_distinct, then per monomial: n_factors, (kind, index) x n_factors
    std::vector<uint32_t> coff, boff;
    std::vector<uint64_t> B;       // [C][n_dist]
    size_t n_dist = 0;
};
void compile_gates(const sipp_plonk_circuit* c, const std::vector<uint64_t>& gapow, uint32_t C, uint32_t n_gc, CompiledGates* out) {
    using Factors = std::vector<std::pair<int64_t, int64_t>>;
    struct Mono {
        Factors f;
        std::vector<uint64_t> b;   // per challenge
    };
    std::vector<std::vector<Mono>> per_gate(c->num_gates);
    for (uint32_t g = 0; g < c->num_gates; g++) {
        const sipp_plonk_gate& ga = c->gates[g];
        std::map<Factors, size_t> index;
        std::vector<Mono>& monos = per_gate[g];
        size_t w = ga.prog_offset;
        for (uint32_t j = 0; j < ga.num_constraints; j++) {
            const int64_t nm = c->programs[w++];
            for (int64_t m = 0; m < nm; m++) {
                const uint64_t coef = gl::from_i64(c->programs[w]);
                const int64_t nf = c->programs[w + 1];
                w += 2;
                Factors f;
                for (int64_t q = 0; q < nf; q++, w += 2) f.emplace_back(c->programs[w], c->programs[w + 1]);
                std::sort(f.begin(), f.end());
                auto it = index.find(f);
                if (it == index.end()) {
                    it = index.emplace(f, monos.size()).first;
                    monos.push_back(Mono{f, std::vector<uint64_t>(C, 0)});
                }
                for (uint32_t cc = 0; cc < C; cc++)
                    monos[it->second].b[cc] = gl::add(monos[it->second].b[cc], gl::mul(gapow[(size_t)cc * n_gc + j], coef));
            }
        }
        auto pure = [](const Factors& f) { return !f.empty() && f.front() == f.back(); };   // sorted: all factors equal
        std::stable_sort(monos.begin(), monos.end(), [&](const Mono& x, const Mono& y) {
            const bool px = pure(x.f), py = pure(y.f);
            if (px != py) return px;
            if (px) return std::make_pair(x.f.front(), x.f.size()) < std::make_pair(y.f.front(), y.f.size());
            return false;
        });
        out->n_dist += monos.size();
    }
    out->B.assign((size_t)C * std::max<size_t>(1, out->n_dist), 0);
    const size_t stride = std::max<size_t>(1, out->n_dist);
    size_t base = 0;
    for (uint32_t g = 0; g < c->num_gates; g++) {
        out->coff.push_back((uint32_t)out->prog.size());
        out->boff.push_back((uint32_t)base);
        out->prog.push_back((int64_t)per_gate[g].size());
        for (const Mono& m : per_gate[g]) {
            out->prog.push_back((int64_t)m.f.size());
            for (const auto& kv : m.f) {
                out->prog.push_back(kv.first);
                out->prog.push_back(kv.second);
            }
            for (uint32_t cc = 0; cc < C; cc++) out->B[(size_t)cc * stride + base] = m.b[cc];
            base++;
        }
    }
}

uint32_t num_gate_constraints(const sipp_plonk_circuit* c) {
    uint32_t m = 0;
    for (uint32_t g = 0; g < c->num_gates; g++) m = std::max(m, c->gates[g].num_constraints);
    return m;
}

// every operand in range, every program inside program_words: an index out of range would be a read outside the LDE buffers
int circuit_check(sipp_ctx* ctx, const sipp_plonk_circuit* c, const sipp_plonk_params* p) {
    if (!c || !p || c->num_wires < p->num_routed_wires || c->num_wires > 4096 || c->num_selectors == 0 || c->num_selectors > c->num_constants ||
        c->num_constants > 1024 || c->num_gates == 0 || c->num_gates > 4096 || !c->gates || (!c->programs && c->program_words))
        return sipp_fail(ctx, SIPP_E_BADARG, "plonk: malformed circuit description");
    for (uint32_t g = 0; g < c->num_gates; g++) {
        const sipp_plonk_gate& ga = c->gates[g];
        if (ga.selector_index >= c->num_selectors || ga.group_lo > ga.row || ga.row >= ga.group_hi || ga.group_hi > c->num_gates ||
            ga.group_hi - ga.group_lo > 64 || ga.num_constraints > 4096)
            return sipp_fail(ctx, SIPP_E_BADARG, "plonk: a gate's selector group is malformed");
        size_t w = ga.prog_offset;
        for (uint32_t j = 0; j < ga.num_constraints; j++) {
            if (w >= c->program_words) return sipp_fail(ctx, SIPP_E_BADARG, "plonk: a gate program runs past program_words");
            const int64_t nm = c->programs[w++];
            if (nm < 0 || nm > 4096) return sipp_fail(ctx, SIPP_E_BADARG, "plonk: a gate program is malformed");
            for (int64_t m = 0; m < nm; m++) {
                if (w + 2 > c->program_words) return sipp_fail(ctx, SIPP_E_BADARG, "plonk: a gate program runs past program_words");
                const int64_t nf = c->programs[w + 1];
                w += 2;
                if (nf < 0 || nf > 64 || w + 2 * (size_t)nf > c->program_words)
                    return sipp_fail(ctx, SIPP_E_BADARG, "plonk: a gate program runs past program_words");
                for (int64_t f = 0; f < nf; f++, w += 2) {
                    const int64_t kind = c->programs[w], idx = c->programs[w + 1];
                    if (kind < 0 || kind > 2 || idx < 0 || (kind == 0 && idx >= c->num_wires) || (kind == 1 && idx >= c->num_constants) ||
                        (kind == 2 && idx >= 4))
                        return sipp_fail(ctx, SIPP_E_BADARG, "plonk: a gate program names an operand out of range");
                }
            }
        }
    }
    return SIPP_OK;
}

int quotient_chunks_impl(sipp_ctx* ctx, const uint64_t* d_wires_lde, const uint64_t* d_sigmas_lde, const uint64_t* d_zs_lde, uint32_t log_n,
                         uint32_t rate_bits, const sipp_plonk_params* p, const uint64_t* betas, const uint64_t* gammas, const uint64_t* alphas,
                         const uint64_t* d_gate_terms, uint32_t num_gate_terms, const GateSet* gs, uint64_t* d_chunks) {
    if ((num_gate_terms != 0) != (d_gate_terms != nullptr) || num_gate_terms > (1u << 20)) return ctx ? sipp_fail(ctx, SIPP_E_BADARG, "plonk: gate terms and their count disagree") : SIPP_E_BADARG;

    if (!ctx || !d_wires_lde || !d_sigmas_lde || !d_zs_lde || !betas || !gammas || !alphas || !d_chunks) return SIPP_E_BADARG;
    uint32_t log_d, m;
    SIPP_TRY(check(ctx, p, log_n, &log_d, &m));
    if (rate_bits < log_d || rate_bits > 3) return sipp_fail(ctx, SIPP_E_UNSUPPORTED, "plonk: the blowup must cover the quotient degree factor (and be <= 8)");
    SIPP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    ArenaScope scope(ctx);
    const size_t n = (size_t)1 << log_n, nd = n << log_d;
    const uint32_t C = p->num_challenges;
    QuotArgs a{};
    a.wl = d_wires_lde; a.sl = d_sigmas_lde; a.zl = d_zs_lde; a.stride = n << rate_bits;
    a.log_n = log_n; a.rate_bits = rate_bits; a.log_d = log_d; a.R = p->num_routed_wires; a.D = p->max_degree; a.m = m; a.C = C;
    for (uint32_t c = 0; c < C; c++) { a.beta[c] = gl::canon(betas[c]); a.gamma[c] = gl::canon(gammas[c]); a.alpha[c] = gl::canon(alphas[c]); }
    a.w_nd = gl::root_of_unity(log_n + log_d);
    a.n_field = (uint64_t)n;
    {   // x^n = 7^n w_D^(i mod D) on the coset
        const uint64_t g_n = gl::pow(gl::GEN, (uint64_t)n), w_d = gl::root_of_unity(log_d);
        uint64_t f = g_n;
        for (uint32_t d = 0; d < p->max_degree; d++) {
            a.zh[d] = gl::sub(f, 1);
            a.zh_inv[d] = gl::inv(a.zh[d]);
            f = gl::mul(f, w_d);
        }
    }
    a.bk = upload_bk(ctx, p, betas);
    if (!a.bk) return SIPP_E_NOMEM;
    a.qv = d_chunks;           // [C][n D]: values in leaf order, transformed in place
    a.gt = d_gate_terms;
    a.n_gt = num_gate_terms;
    if (gs) {
        const sipp_plonk_circuit* c = gs->c;
        a.cl = gs->d_consts_lde;
        a.n_gates = c->num_gates; a.many_sel = c->num_selectors > 1;
        const uint32_t n_gc = num_gate_constraints(c);
        for (int q = 0; q < 4; q++) a.pih[q] = gl::canon(gs->pih[q]);
        // alpha_c^(terms in front of gate constraint j): the permutation terms C + C m, then the caller's own terms
        std::vector<uint64_t> gp((size_t)C * std::max(1u, n_gc));
        for (uint32_t cc = 0; cc < C; cc++) {
            uint64_t x = gl::pow(a.alpha[cc], (uint64_t)C + (uint64_t)C * m + num_gate_terms);
            for (uint32_t j = 0; j < n_gc; j++) {
                gp[(size_t)cc * n_gc + j] = x;
                x = gl::mul(x, a.alpha[cc]);
            }
        }
        CompiledGates cg;
        compile_gates(c, gp, C, n_gc, &cg);
        a.n_dist = (uint32_t)std::max<size_t>(1, cg.n_dist);
        sipp_plonk_gate* d_gates = arena_alloc_t<sipp_plonk_gate>(ctx, c->num_gates);
        int64_t* d_cprog = arena_alloc_t<int64_t>(ctx, std::max<size_t>(1, cg.prog.size()));
        uint32_t* d_off = arena_alloc_t<uint32_t>(ctx, 2 * (size_t)c->num_gates);
        uint64_t* d_B = arena_alloc_t<uint64_t>(ctx, std::max<size_t>(1, cg.B.size()));
        if (!d_gates || !d_cprog || !d_off || !d_B) return SIPP_E_NOMEM;
        std::vector<uint32_t> off(cg.coff);
        off.insert(off.end(), cg.boff.begin(), cg.boff.end());
        SIPP_CHECK_HIP(ctx, hipMemcpyAsync(d_gates, c->gates, (size_t)c->num_gates * sizeof(sipp_plonk_gate), hipMemcpyHostToDevice, ctx->stream));
        if (!cg.prog.empty()) SIPP_CHECK_HIP(ctx, hipMemcpyAsync(d_cprog, cg.prog.data(), cg.prog.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        SIPP_CHECK_HIP(ctx, hipMemcpyAsync(d_off, off.data(), off.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        if (!cg.B.empty()) SIPP_CHECK_HIP(ctx, hipMemcpyAsync(d_B, cg.B.data(), cg.B.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        SIPP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));      // the vectors above and the caller's arrays are host memory
        a.gates = d_gates; a.cprog = d_cprog; a.coff = d_off; a.boff = d_off + c->num_gates; a.cB = d_B;
    }
    {
        ProfScope ps(ctx, "plonk_quotient");
        hipLaunchKernelGGL(plonk_quotient_kernel, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, ctx->stream, a);
    }
    SIPP_CHECK_HIP(ctx, hipGetLastError());
    // coset_ifft(7): leaf-order values -> natural coefficients; [C][n D] natural == [C D][n] chunks
    SIPP_TRY(sipp_ntt_dit(ctx, d_chunks, nd, log_n + log_d, C, true, NttDiag{gl::inv(gl::GEN), 0}));
    return sipp_sync(ctx);
}
}  // namespace

// circuit_data.hip refuses a malformed gate set at build time
int sipp_plonk_circuit_check(sipp_ctx* ctx, const sipp_plonk_circuit* c, const sipp_plonk_params* p) { return circuit_check(ctx, c, p); }

// the flow of oracle/plonk.c::orc_plonk_perm_prove on the device: four PolynomialBatch commitments, the transcript on the host, one
// opening proof at zeta / g zeta.  Flat proof: header[8] | wires cap | zs_partial_products cap | quotient cap | opening proof.
namespace {
// Optional parts of the flow: gate-constraint terms in the quotient, public inputs in the transcript and behind the proof ("SIPPPLK2"),
// and oracles the caller committed already (sigmas once per circuit; wires before it evaluated its gates on their LDE)
struct PlonkExtra {
    const uint64_t* d_gate_terms = nullptr;
    uint32_t n_gate_terms = 0;
    const uint64_t* public_inputs = nullptr;
    uint32_t n_public_inputs = 0;
    bool v2 = false;
    const sipp_oracle* sigmas_oracle = nullptr;
    const sipp_oracle* wires_oracle = nullptr;
    const uint64_t* wires_cap = nullptr;
    // gates as data ("SIPPPLK3"): d_wires holds circ->num_wires columns, d_sigmas is the WHOLE constants_sigmas batch (num_constants
    // columns in front of the sigmas)
    const sipp_plonk_circuit* circ = nullptr;
};
int plonk_prove_impl(sipp_ctx* ctx, const uint64_t* d_wires, const uint64_t* d_sigmas, uint32_t log_n, const sipp_plonk_params* p,
                     const sipp_fri_params* fp, const uint64_t circuit_digest[4], const uint64_t public_inputs_hash[4], const PlonkExtra& ex,
                     uint64_t* proof_out, size_t proof_cap, size_t* proof_len);
}  // namespace

int sipp_plonk_perm_prove(sipp_ctx* ctx, const uint64_t* d_wires, const uint64_t* d_sigmas, uint32_t log_n, const sipp_plonk_params* p,
                          const sipp_fri_params* fp, const uint64_t circuit_digest[4], const uint64_t public_inputs_hash[4],
                          uint64_t* proof_out, size_t proof_cap, size_t* proof_len) {
    if (!ctx || !d_wires || !d_sigmas || !fp || !circuit_digest || !public_inputs_hash || !proof_out || !proof_len) return SIPP_E_BADARG;
    return plonk_prove_impl(ctx, d_wires, d_sigmas, log_n, p, fp, circuit_digest, public_inputs_hash, PlonkExtra{}, proof_out, proof_cap, proof_len);
}

int sipp_plonk_prove_ex(sipp_ctx* ctx, const uint64_t* d_wires, const uint64_t* d_sigmas, const sipp_oracle* wires_oracle, const uint64_t* wires_cap,
                        const sipp_oracle* sigmas_oracle, uint32_t log_n, const sipp_plonk_params* p, const sipp_fri_params* fp,
                        const uint64_t circuit_digest[4], const uint64_t* public_inputs, uint32_t n_public_inputs, const uint64_t* d_gate_terms,
                        uint32_t num_gate_terms, uint64_t* proof_out, size_t proof_cap, size_t* proof_len) {
    if (!ctx || !d_wires || !d_sigmas || !fp || !p || !circuit_digest || !proof_out || !proof_len || (n_public_inputs && !public_inputs) ||
        (num_gate_terms != 0) != (d_gate_terms != nullptr) || (wires_oracle != nullptr) != (wires_cap != nullptr) || n_public_inputs > (1u << 24))
        return SIPP_E_BADARG;
    if ((wires_oracle && (wires_oracle->n_polys != p->num_routed_wires || wires_oracle->n_salt || !wires_oracle->d_coeffs || !wires_oracle->d_lde ||
                          !wires_oracle->d_tree)) ||
        (sigmas_oracle && (sigmas_oracle->n_polys != p->num_routed_wires || sigmas_oracle->n_salt || !sigmas_oracle->d_coeffs ||
                           !sigmas_oracle->d_lde || !sigmas_oracle->d_tree)))
        return sipp_fail(ctx, SIPP_E_BADARG, "plonk: a pre-committed oracle must hold num_routed_wires unsalted polynomials");
    uint64_t pih[4];
    host::Challenger::hash_no_pad(public_inputs, n_public_inputs, pih);      // plonk/prover.rs: hash_n_to_hash_no_pad(public_inputs)
    PlonkExtra ex;
    ex.d_gate_terms = d_gate_terms; ex.n_gate_terms = num_gate_terms; ex.public_inputs = public_inputs; ex.n_public_inputs = n_public_inputs;
    ex.v2 = true; ex.sigmas_oracle = sigmas_oracle; ex.wires_oracle = wires_oracle; ex.wires_cap = wires_cap;
    return plonk_prove_impl(ctx, d_wires, d_sigmas, log_n, p, fp, circuit_digest, pih, ex, proof_out, proof_cap, proof_len);
}

namespace {
int plonk_prove_impl(sipp_ctx* ctx, const uint64_t* d_wires, const uint64_t* d_sigmas, uint32_t log_n, const sipp_plonk_params* p,
                     const sipp_fri_params* fp, const uint64_t circuit_digest[4], const uint64_t public_inputs_hash[4], const PlonkExtra& ex,
                     uint64_t* proof_out, size_t proof_cap, size_t* proof_len) {
    uint32_t log_d, m;
    SIPP_TRY(check(ctx, p, log_n, &log_d, &m));
    if (fp->rate_bits < log_d || fp->rate_bits > 3 || fp->cap_height > 8 || fp->hiding)
        return sipp_fail(ctx, SIPP_E_UNSUPPORTED, "plonk: blowup 2^rate_bits >= quotient degree factor, <= 8, unsalted oracles");
    SIPP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    ArenaScope scope(ctx);
    const uint32_t R = p->num_routed_wires, D = p->max_degree, C = p->num_challenges, nz = C * m;
    const size_t n = (size_t)1 << log_n, M = n << fp->rate_bits, cap_n = (size_t)1 << std::min(fp->cap_height, log_n + fp->rate_bits);
    const uint32_t K = ex.circ ? ex.circ->num_constants : 0, Wn = ex.circ ? ex.circ->num_wires : R;
    const uint32_t ncols[4] = {K + R, Wn, nz, C * D};
    const uint64_t* d_sigma_vals = d_sigmas + (size_t)K * n;      // the sigmas behind the constant columns
    uint64_t *co[4], *lde[4], *tree[4];
    const sipp_oracle* pre[4] = {ex.sigmas_oracle, ex.wires_oracle, nullptr, nullptr};
    for (int o = 0; o < 4; o++) {
        if (pre[o]) {      // committed by the caller: read only
            co[o] = const_cast<uint64_t*>(pre[o]->d_coeffs);
            lde[o] = const_cast<uint64_t*>(pre[o]->d_lde);
            tree[o] = const_cast<uint64_t*>(pre[o]->d_tree);
            continue;
        }
        co[o] = arena_alloc_t<uint64_t>(ctx, (size_t)ncols[o] * n);
        lde[o] = arena_alloc_t<uint64_t>(ctx, (size_t)ncols[o] * M);
        tree[o] = arena_alloc_t<uint64_t>(ctx, 2 * M * 4);
        if (!co[o] || !lde[o] || !tree[o]) return SIPP_E_NOMEM;
    }
    std::vector<uint64_t> caps(4 * cap_n * 4);
    auto cap_of = [&](int o) { return caps.data() + (size_t)o * cap_n * 4; };
    if (!pre[0])
        SIPP_TRY(sipp_commit_batch_ex(ctx, d_sigmas, 0, co[0], lde[0], tree[0], ncols[0], log_n, fp->rate_bits, fp->cap_height, nullptr, 0, cap_of(0)));
    if (!pre[1])
        SIPP_TRY(sipp_commit_batch_ex(ctx, d_wires, 0, co[1], lde[1], tree[1], ncols[1], log_n, fp->rate_bits, fp->cap_height, nullptr, 0, cap_of(1)));
    else
        memcpy(cap_of(1), ex.wires_cap, cap_n * 32);
    host::Challenger ch;
    ch.observe_many(circuit_digest, 4);
    ch.observe_many(public_inputs_hash, 4);
    ch.observe_many(cap_of(1), cap_n * 4);
    uint64_t betas[MAX_CH], gammas[MAX_CH], alphas[MAX_CH];
    for (uint32_t c = 0; c < C; c++) betas[c] = ch.get();
    for (uint32_t c = 0; c < C; c++) gammas[c] = ch.get();
    {
        uint64_t* zs = arena_alloc_t<uint64_t>(ctx, (size_t)nz * n);
        if (!zs) return SIPP_E_NOMEM;
        SIPP_TRY(sipp_plonk_zs_partial_products(ctx, d_wires, d_sigma_vals, log_n, p, betas, gammas, zs));
        SIPP_TRY(sipp_commit_batch_ex(ctx, zs, 0, co[2], lde[2], tree[2], nz, log_n, fp->rate_bits, fp->cap_height, nullptr, 0, cap_of(2)));
    }
    ch.observe_many(cap_of(2), cap_n * 4);
    for (uint32_t c = 0; c < C; c++) alphas[c] = ch.get();
    {
        const GateSet gs{lde[0], ex.circ, public_inputs_hash};
        SIPP_TRY(quotient_chunks_impl(ctx, lde[1], lde[0] + (size_t)K * M, lde[2], log_n, fp->rate_bits, p, betas, gammas, alphas, ex.d_gate_terms,
                                      ex.n_gate_terms, ex.circ ? &gs : nullptr, co[3]));
    }
    SIPP_TRY(sipp_commit_batch_ex(ctx, co[3], 1, co[3], lde[3], tree[3], (size_t)C * D, log_n, fp->rate_bits, fp->cap_height, nullptr, 0, cap_of(3)));
    ch.observe_many(cap_of(3), cap_n * 4);
    const gl::E2 zeta = ch.get_ext();
    sipp_oracle oracles[4];
    for (int o = 0; o < 4; o++) oracles[o] = sipp_oracle{co[o], lde[o], tree[o], ncols[o], 0};
    const sipp_poly_range r0[4] = {{0, 0, ncols[0]}, {1, 0, ncols[1]}, {2, 0, nz}, {3, 0, C * D}}, r1[1] = {{2, 0, C}};
    const gl::E2 gz = gl::scale(zeta, gl::root_of_unity(log_n));
    sipp_fri_batch batches[2] = {{{zeta.c0, zeta.c1}, 4, r0}, {{gz.c0, gz.c1}, 1, r1}};
    const size_t op_cap = sipp_fri_proof_size(oracles, 4, batches, 2, log_n, fp);
    const size_t hw = ex.circ ? 16 : 8;                          // header words
    const size_t head = hw + 3 * cap_n * 4, tail = ex.v2 ? ex.n_public_inputs : 0;
    if (op_cap == 0) return sipp_fail(ctx, SIPP_E_BADARG, "plonk: FRI parameters do not fit the degree");
    if (proof_cap < head + op_cap + tail) return sipp_fail(ctx, SIPP_E_BUFSZ, "plonk: proof buffer too small (see sipp_plonk_perm_proof_size)");
    sipp_challenger cs{};
    memcpy(cs.state, ch.state, sizeof cs.state);
    memcpy(cs.in_buf, ch.in_buf, sizeof cs.in_buf);
    memcpy(cs.out_buf, ch.out_buf, sizeof cs.out_buf);
    cs.n_in = ch.n_in;
    cs.n_out = ch.n_out;
    size_t op_len = 0;
    SIPP_TRY(sipp_fri_prove_openings(ctx, oracles, 4, batches, 2, log_n, fp, &cs, proof_out + head, proof_cap - head - tail, &op_len));
    if (ex.circ) {
        const uint64_t h[16] = {0x334b4c5050504953ULL /* "SIPPPLK3" */, log_n, R, D, C, head + op_len + tail, Wn, K, ex.circ->num_selectors,
                                ex.circ->num_gates, num_gate_constraints(ex.circ), tail, 0, 0, 0, 0};
        memcpy(proof_out, h, sizeof h);
    } else {
        const uint64_t h[8] = {ex.v2 ? 0x324b4c5050504953ULL /* "SIPPPLK2" */ : 0x314b4c5050504953ULL /* "SIPPPLK1" */, log_n, R, D, C,
                               head + op_len + tail, ex.v2 ? ex.n_gate_terms : 0, tail};
        memcpy(proof_out, h, sizeof h);
    }
    if (tail) memcpy(proof_out + head + op_len, ex.public_inputs, tail * 8);
    memcpy(proof_out + hw, cap_of(1), cap_n * 32);
    memcpy(proof_out + hw + cap_n * 4, cap_of(2), cap_n * 32);
    memcpy(proof_out + hw + 2 * cap_n * 4, cap_of(3), cap_n * 32);
    *proof_len = head + op_len + tail;
    return SIPP_OK;
}
}  // namespace

size_t sipp_plonk_perm_proof_size(uint32_t log_n, const sipp_plonk_params* p, const sipp_fri_params* fp) {
    // the bounds of check(): sizes below are u32 products
    if (!p || !fp || p->max_degree < 2 || p->max_degree > 64 || (p->max_degree & (p->max_degree - 1)) || p->num_routed_wires == 0 ||
        p->num_challenges == 0 || p->num_challenges > MAX_CH || (p->num_routed_wires + p->max_degree - 1) / p->max_degree > MAX_CHUNKS ||
        log_n < 1 || log_n > 24)
        return 0;
    const uint32_t R = p->num_routed_wires, D = p->max_degree, C = p->num_challenges, nz = C * ((R + D - 1) / D);
    const size_t cap_n = (size_t)1 << std::min(fp->cap_height, log_n + fp->rate_bits);
    sipp_oracle oracles[4] = {{nullptr, nullptr, nullptr, R, 0}, {nullptr, nullptr, nullptr, R, 0}, {nullptr, nullptr, nullptr, nz, 0},
                              {nullptr, nullptr, nullptr, C * D, 0}};
    const sipp_poly_range r0[4] = {{0, 0, R}, {1, 0, R}, {2, 0, nz}, {3, 0, C * D}}, r1[1] = {{2, 0, C}};
    sipp_fri_batch batches[2] = {{{0, 0}, 4, r0}, {{0, 0}, 1, r1}};
    const size_t op = sipp_fri_proof_size(oracles, 4, batches, 2, log_n, fp);
    return op ? 8 + 3 * cap_n * 4 + op : 0;
}

size_t sipp_plonk_gates_proof_size(uint32_t log_n, const sipp_plonk_params* p, const sipp_fri_params* fp, const sipp_plonk_circuit* c,
                                   uint32_t n_public_inputs) {
    if (!p || !fp || !c || p->max_degree < 2 || p->max_degree > 64 || (p->max_degree & (p->max_degree - 1)) || p->num_routed_wires == 0 ||
        p->num_challenges == 0 || p->num_challenges > MAX_CH || (p->num_routed_wires + p->max_degree - 1) / p->max_degree > MAX_CHUNKS ||
        log_n < 1 || log_n > 24 || c->num_wires < p->num_routed_wires || c->num_wires > 4096 || c->num_constants > 1024)
        return 0;
    const uint32_t R = p->num_routed_wires, D = p->max_degree, C = p->num_challenges, nz = C * ((R + D - 1) / D);
    const size_t cap_n = (size_t)1 << std::min(fp->cap_height, log_n + fp->rate_bits);
    sipp_oracle oracles[4] = {{nullptr, nullptr, nullptr, c->num_constants + R, 0}, {nullptr, nullptr, nullptr, c->num_wires, 0},
                              {nullptr, nullptr, nullptr, nz, 0}, {nullptr, nullptr, nullptr, C * D, 0}};
    const sipp_poly_range r0[4] = {{0, 0, c->num_constants + R}, {1, 0, c->num_wires}, {2, 0, nz}, {3, 0, C * D}}, r1[1] = {{2, 0, C}};
    sipp_fri_batch batches[2] = {{{0, 0}, 4, r0}, {{0, 0}, 1, r1}};
    const size_t op = sipp_fri_proof_size(oracles, 4, batches, 2, log_n, fp);
    return op ? 16 + 3 * cap_n * 4 + op + n_public_inputs : 0;
}

int sipp_plonk_prove_gates(sipp_ctx* ctx, const uint64_t* d_wires, const uint64_t* d_constants_sigmas, const sipp_oracle* wires_oracle,
                           const uint64_t* wires_cap, const sipp_oracle* constants_sigmas_oracle, uint32_t log_n, const sipp_plonk_params* p,
                           const sipp_fri_params* fp, const sipp_plonk_circuit* c, const uint64_t circuit_digest[4], const uint64_t* public_inputs,
                           uint32_t n_public_inputs, uint64_t* proof_out, size_t proof_cap, size_t* proof_len) {
    if (!ctx || !d_wires || !d_constants_sigmas || !fp || !p || !c || !circuit_digest || !proof_out || !proof_len ||
        (n_public_inputs && !public_inputs) || (wires_oracle != nullptr) != (wires_cap != nullptr) || n_public_inputs > (1u << 24))
        return SIPP_E_BADARG;
    SIPP_TRY(circuit_check(ctx, c, p));
    if ((wires_oracle && (wires_oracle->n_polys != c->num_wires || wires_oracle->n_salt || !wires_oracle->d_coeffs || !wires_oracle->d_lde ||
                          !wires_oracle->d_tree)) ||
        (constants_sigmas_oracle &&
         (constants_sigmas_oracle->n_polys != c->num_constants + p->num_routed_wires || constants_sigmas_oracle->n_salt ||
          !constants_sigmas_oracle->d_coeffs || !constants_sigmas_oracle->d_lde || !constants_sigmas_oracle->d_tree)))
        return sipp_fail(ctx, SIPP_E_BADARG, "plonk: a pre-committed oracle must hold every column of its batch, unsalted");
    uint64_t pih[4];
    host::Challenger::hash_no_pad(public_inputs, n_public_inputs, pih);
    PlonkExtra ex;
    ex.public_inputs = public_inputs; ex.n_public_inputs = n_public_inputs; ex.v2 = true;
    ex.sigmas_oracle = constants_sigmas_oracle; ex.wires_oracle = wires_oracle; ex.wires_cap = wires_cap; ex.circ = c;
    return plonk_prove_impl(ctx, d_wires, d_constants_sigmas, log_n, p, fp, circuit_digest, pih, ex, proof_out, proof_cap, proof_len);
}
