// sipp_amd/csrc/stark.hip -- host orchestration of the three provers behind reference
// src/verifier_circuit.rs:133-135 (g1_exp_circuit / g2_exp_circuit / fq12_exp_circuit): the body of
// starky-bn254's proof generators, i.e. generate_trace + starky::prover::prove, on one MI355X.
//
// Sequence (flat proof layout documented in INTEGRATION.md and oracle/stark.c):
//   trace fill -> commit (iNTT, coset LDE, Poseidon Merkle) -> (beta, gamma) -> Z columns -> commit
//   -> alphas -> quotient on the 2N coset -> coset iNTT -> 4 chunks -> commit -> zeta -> openings
//   -> FRI: alpha-combine, divide by (X - z), fold x16 per round with a Merkle commit, final poly,
//      proof-of-work (smallest nonce), 84 query openings.
// The Fiat-Shamir challenger runs on the host (a few hundred Poseidon permutations per proof); every
// array of size O(N) stays in HBM.  One HIP stream per ctx; three ctxs give the three-stream overlap.
#include <atomic>
#include <thread>
#include <algorithm>
#include <chrono>

#include "prover.hpp"

static const int IO_WORDS[7] = {SIPP_G1_IO_WORDS, SIPP_G2_IO_WORDS, SIPP_FQ12_IO_WORDS, SIPP_MAP_G2_IO_WORDS, SIPP_G1_IO_WORDS, SIPP_G2_IO_WORDS,
                                SIPP_PAIRING_IO_WORDS};
// the hardened G1 / G2 kinds take the records of the plain ones
static inline int base_kind(int kind) {
    return (kind == SIPP_G1_EXP_HARDENED || kind == SIPP_G2_EXP_HARDENED) ? kind - SIPP_G1_EXP_HARDENED : kind;
}
// sipp_ctx_set_hardened: on such a ctx the plain G1 / G2 kinds stand for the hardened ones (every entry point that takes a ctx)
static inline int ctx_kind(const sipp_ctx* ctx, int kind) {
    return (ctx && ctx->hardened && (kind == SIPP_G1_EXP || kind == SIPP_G2_EXP)) ? kind + SIPP_G1_EXP_HARDENED : kind;
}
#define SIPP_MAGIC 0x5349505053544b31ULL /* "SIPPSTK1" */

struct Shape {
    const air_spec_t* air;
    uint32_t log_n;
    uint32_t num_io;  // padded
    int W, P, Q;
};

static int shape_of(int kind, size_t num_io, Shape* s) {
    if (kind < 0 || kind > SIPP_PAIRING || num_io == 0 || num_io > ((size_t)1 << 17)) return SIPP_E_BADARG;
    // rows per record: 512 (exponentiations), 8 (MapToG2), 8192 (the final pairing)
    const uint32_t log_rows = kind == SIPP_MAP_G2 ? 3 : kind == SIPP_PAIRING ? AIR_PAIRING_LOG_ROWS : 9;
    uint32_t nio = 2;  // at least two IO blocks, at least 1024 rows
    while (nio < num_io || ((size_t)nio << log_rows) < 1024) nio <<= 1;
    uint32_t log_n = log_rows;
    while ((1u << (log_n - log_rows)) < nio) log_n++;
    s->air = sipp_air_get(kind, log_n);
    if (!s->air || (uint32_t)s->air->log_rows != log_rows) return SIPP_E_UNSUPPORTED;
    s->log_n = log_n;
    s->num_io = nio;
    s->W = s->air->n_main + 2 * s->air->n_checked;
    s->P = 2 * s->air->n_checked;
    s->Q = 4;
    return SIPP_OK;
}

// FriParams of a STARK: FriReductionStrategy::ConstantArityBits(arity_bits, final_poly_bits)
static FriParamsDev fri_params_of(const sipp_stark_config& c, uint32_t degree_bits) {
    FriParamsDev p;
    p.rate_bits = c.rate_bits; p.cap_height = c.cap_height; p.pow_bits = c.pow_bits; p.num_queries = c.num_queries;
    p.pow_rule = c.pow_rule;
    while (degree_bits > c.final_poly_bits && degree_bits + c.rate_bits - c.arity_bits >= c.cap_height &&
           degree_bits >= c.arity_bits && p.arity_bits.size() < 32) {
        p.arity_bits.push_back(c.arity_bits);
        degree_bits -= c.arity_bits;
    }
    return p;
}

static size_t proof_words(const sipp_stark_config& cfg, const Shape& s) {
    const FriParamsDev fp = fri_params_of(cfg, s.log_n);
    const uint32_t log_m = s.log_n + cfg.rate_bits;
    const size_t cap = (size_t)4 << cfg.cap_height;
    size_t w = 16 + 3 * cap + 2 * (size_t)(2 * s.W + 2 * s.P + s.Q);
    const uint32_t leaf_words[3] = {(uint32_t)s.W, (uint32_t)s.P, (uint32_t)s.Q};
    w += sipp_fri_core_words(fp, s.log_n, leaf_words, 3);
    w += (size_t)s.num_io * s.air->pi_per_io;
    (void)log_m;
    return w;
}

// pads the IO list with copies of the last record and uploads it
static int upload_ios(sipp_ctx* ctx, int kind, const uint32_t* ios, size_t num_io, const Shape& s, uint32_t** d_ios,
                      std::vector<uint32_t>* host_copy) {
    const size_t ppi = IO_WORDS[kind];
    const size_t words = (size_t)s.num_io * ppi;
    if (words * 4 > ctx->h_pinned_words * 8) return sipp_fail(ctx, SIPP_E_NOMEM, "IO list larger than the pinned staging buffer");
    uint32_t* h = reinterpret_cast<uint32_t*>(ctx->h_pinned);
    for (size_t io = 0; io < s.num_io; io++)
        memcpy(h + io * ppi, ios + (io < num_io ? io : num_io - 1) * ppi, ppi * 4);
    if (host_copy) host_copy->assign(h, h + words);
    *d_ios = arena_alloc_t<uint32_t>(ctx, words);
    if (!*d_ios) return SIPP_E_NOMEM;
    SIPP_CHECK_HIP(ctx, hipMemcpyAsync(*d_ios, h, words * 4, hipMemcpyHostToDevice, ctx->stream));
    SIPP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the pinned buffer is reused below
    return SIPP_OK;
}

// ---- commitment helpers -------------------------------------------------------------------------------
static size_t tree_words(uint32_t log_leaves) { return ((size_t)8 << log_leaves); }  // 2 * leaves * 4

static int read_cap(sipp_ctx* ctx, const uint64_t* d_tree, uint32_t log_leaves, uint64_t* cap_host) {
    const uint32_t ch = std::min(ctx->cfg.cap_height, log_leaves);
    uint64_t off = 0;
    for (uint32_t l = 0; l < log_leaves - ch; l++) off += (uint64_t)1 << (log_leaves - l);
    SIPP_CHECK_HIP(ctx, hipMemcpyAsync(cap_host, d_tree + 4 * off, ((size_t)4 << ch) * 8, hipMemcpyDeviceToHost, ctx->stream));
    SIPP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SIPP_OK;
}

// coeffs [ncols][n] natural -> lde [ncols][m] leaf order -> tree (launches only; read_cap collects the cap)
static int commit_coeffs_launch(sipp_ctx* ctx, const uint64_t* d_coeffs, size_t ncols, uint32_t log_n, uint64_t* d_lde,
                                uint64_t* d_tree) {
    const uint32_t log_m = log_n + ctx->cfg.rate_bits;
    int rc = sipp_lde_from_coeffs(ctx, d_coeffs, d_lde, ncols, log_n, ctx->cfg.rate_bits);
    if (rc == SIPP_E_UNSUPPORTED)
        rc = sipp_ntt_dif(ctx, d_coeffs, (size_t)1 << log_n, log_n, d_lde, (size_t)1 << log_m, log_m, ncols, false,
                          NttDiag{gl::GEN, 0});
    SIPP_TRY(rc);
    SIPP_TRY(sipp_k_poseidon_leaves(ctx, d_lde, (size_t)1 << log_m, ncols, log_m, d_tree));
    return sipp_k_merkle_levels(ctx, d_tree, log_m, ctx->cfg.cap_height);
}
static int commit_coeffs(sipp_ctx* ctx, const uint64_t* d_coeffs, size_t ncols, uint32_t log_n, uint64_t* d_lde,
                         uint64_t* d_tree, uint64_t* cap_host) {
    SIPP_TRY(commit_coeffs_launch(ctx, d_coeffs, ncols, log_n, d_lde, d_tree));
    return read_cap(ctx, d_tree, log_n + ctx->cfg.rate_bits, cap_host);
}

// values [ncols][n] natural -> coeffs -> lde -> tree (launches only)
static int commit_values_launch(sipp_ctx* ctx, const uint64_t* d_values, size_t ncols, uint32_t log_n, uint64_t* d_coeffs,
                                uint64_t* d_lde, uint64_t* d_tree) {
    const size_t n = (size_t)1 << log_n;
    const uint32_t log_m = log_n + ctx->cfg.rate_bits;
    const int rc = sipp_lde_from_values(ctx, d_values, d_coeffs, d_lde, ncols, log_n, ctx->cfg.rate_bits);
    if (rc == SIPP_OK) {   // fused: coefficients and LDE are both in place
        SIPP_TRY(sipp_k_poseidon_leaves(ctx, d_lde, (size_t)1 << log_m, ncols, log_m, d_tree));
        return sipp_k_merkle_levels(ctx, d_tree, log_m, ctx->cfg.cap_height);
    }
    if (rc != SIPP_E_UNSUPPORTED) return rc;
    SIPP_TRY(sipp_bitrev_cols(ctx, d_values, n, d_coeffs, n, log_n, ncols));
    SIPP_TRY(sipp_ntt_dit(ctx, d_coeffs, n, log_n, ncols, true, NttDiag{}));
    return commit_coeffs_launch(ctx, d_coeffs, ncols, log_n, d_lde, d_tree);
}
static int commit_values(sipp_ctx* ctx, const uint64_t* d_values, size_t ncols, uint32_t log_n, uint64_t* d_coeffs,
                         uint64_t* d_lde, uint64_t* d_tree, uint64_t* cap_host) {
    SIPP_TRY(commit_values_launch(ctx, d_values, ncols, log_n, d_coeffs, d_lde, d_tree));
    return read_cap(ctx, d_tree, log_n + ctx->cfg.rate_bits, cap_host);
}

// ---- small host FFT for the public-input polynomials (size = number of IOs) ----------------------------
static void host_ifft(std::vector<uint64_t>& a, uint32_t log_n) {
    const size_t n = (size_t)1 << log_n;
    for (size_t i = 0; i < n; i++) {
        size_t j = gl::bitrev((uint32_t)i, log_n);
        if (i < j) std::swap(a[i], a[j]);
    }
    const uint64_t root = gl::inv(gl::root_of_unity(log_n));
    for (uint32_t s = 1; s <= log_n; s++) {
        const size_t mlen = (size_t)1 << s, h = mlen >> 1;
        uint64_t wm = root;
        for (uint32_t k = s; k < log_n; k++) wm = gl::sqr(wm);
        for (size_t k = 0; k < n; k += mlen) {
            uint64_t w = 1;
            for (size_t j = 0; j < h; j++) {
                uint64_t t = gl::mul(w, a[k + j + h]), u = a[k + j];
                a[k + j] = gl::add(u, t);
                a[k + j + h] = gl::sub(u, t);
                w = gl::mul(w, wm);
            }
        }
    }
    const uint64_t ninv = gl::inv((uint64_t)n);
    for (auto& v : a) v = gl::mul(v, ninv);
}

// ---- BN254 Fq on the host, only for the public basis change of Fq12 public inputs (tools/air_gen.py build_fq12):
// tower component t of the MyFq12 value c[0..11] (8 x u32 LE each): t = 2i -> (c_i + 9 c_{i+6}) mod p, t = 2i+1 -> c_{i+6}
static void fq12_tower_limbs(const uint32_t* c96, int t, uint16_t limbs[16]) {
    static const uint64_t P[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    typedef unsigned __int128 u128;
    const int i = t >> 1;
    uint64_t r[5] = {0, 0, 0, 0, 0};
    auto load = [&](int k, uint64_t out[4]) {
        for (int q = 0; q < 4; q++) out[q] = (uint64_t)c96[8 * k + 2 * q] | ((uint64_t)c96[8 * k + 2 * q + 1] << 32);
    };
    uint64_t hi[4];
    load(i + 6, hi);
    if (t & 1) {
        memcpy(r, hi, 32);
    } else {
        uint64_t lo[4];
        load(i, lo);
        u128 carry = 0;
        for (int q = 0; q < 4; q++) {
            carry += (u128)hi[q] * 9 + lo[q];
            r[q] = (uint64_t)carry;
            carry >>= 64;
        }
        r[4] = (uint64_t)carry;
        // reduce: r < 10 p, subtract p while r >= p
        for (;;) {
            bool ge = r[4] != 0;
            if (!ge) {
                ge = true;
                for (int q = 3; q >= 0; q--) {
                    if (r[q] > P[q]) break;
                    if (r[q] < P[q]) { ge = false; break; }
                }
            }
            if (!ge) break;
            u128 borrow = 0;
            for (int q = 0; q < 4; q++) {
                u128 d = (u128)r[q] - P[q] - borrow;
                r[q] = (uint64_t)d;
                borrow = (d >> 64) & 1;
            }
            r[4] -= (uint64_t)borrow;
        }
    }
    for (int l = 0; l < 16; l++) limbs[l] = (uint16_t)(r[l / 4] >> (16 * (l % 4)));
}

// ---- statement binding (oracle/stark.c header): canonical public inputs, pi_root, the 16 statement words ----------
static bool fq_words_canonical(const uint32_t* w) {
    static const uint64_t P[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    for (int q = 3; q >= 0; q--) {
        const uint64_t v = (uint64_t)w[2 * q] | ((uint64_t)w[2 * q + 1] << 32);
        if (v < P[q]) return true;
        if (v > P[q]) return false;
    }
    return false;
}
// every Fq element of every record < p (the exponent may be any 256-bit value): (x, offset, exp_val, output)
static bool pis_canonical(int kind, const uint32_t* pis, size_t num_io) {
    kind = base_kind(kind);
    if (kind == SIPP_MAP_G2 || kind == SIPP_PAIRING) {   // (u, x, y): six Fq elements; (P, Q, Z): eighteen; no exponent
        for (size_t k = 0; k < (size_t)(kind == SIPP_MAP_G2 ? 6 : 18) * num_io; k++)
            if (!fq_words_canonical(pis + 8 * k)) return false;
        return true;
    }
    const int fe = kind == SIPP_G1_EXP ? 2 : kind == SIPP_G2_EXP ? 4 : 12, ppi = 8 * (3 * fe + 1);
    for (size_t io = 0; io < num_io; io++)
        for (int k = 0; k < 3 * fe + 1; k++)
            if (k != 2 * fe && !fq_words_canonical(pis + io * ppi + 8 * k)) return false;
    return true;
}
// Merkle root (two_to_one) over hash_no_pad(record words); num_io is a power of two >= 2
static void pi_root(const uint32_t* pis, size_t num_io, int ppi, uint64_t root[4]) {
    std::vector<uint64_t> d(num_io * 4), tmp((size_t)ppi);
    for (size_t io = 0; io < num_io; io++) {
        for (int k = 0; k < ppi; k++) tmp[k] = pis[io * ppi + k];
        host::Challenger::hash_no_pad(tmp.data(), (size_t)ppi, &d[4 * io]);
    }
    for (size_t cnt = num_io; cnt > 1; cnt >>= 1)
        for (size_t i = 0; i < cnt / 2; i++) {
            uint64_t out[4];
            host::Challenger::two_to_one(&d[8 * i], &d[8 * i + 4], out);
            memcpy(&d[4 * i], out, 32);
        }
    memcpy(root, d.data(), 32);
}

struct Oracle3 {
    uint64_t *coeffs, *lde, *tree;
    int ncols;
};

static int prove_impl(sipp_ctx* ctx, int kind, const uint32_t* ios, size_t num_io_in, uint64_t* proof_out, size_t proof_cap,
                      size_t* proof_len) {
    if (!ctx || !ios || !proof_out || !proof_len) return SIPP_E_BADARG;
    kind = ctx_kind(ctx, kind);
    struct GateGuard {  // whatever happens below, a proof waiting on this one is let go
        sipp_ctx* c;
        ~GateGuard() {
            if (c->gate_release) c->gate_release->release();
            c->gate_release = nullptr;
        }
    } gate_guard{ctx};
    // where a gated proof waits: inside its trace fill, after the thin doubling / scan chain and before the first wide kernel
    // (trace.hip gate_after_chain; a kind without such a chain waits after its trace fill).  Measured alternatives: before the first
    // launch +0.3 ms per n = 128 instance over five alternating pairs; after the whole trace fill 59.7 against 58.7 ms.
    SIPP_CHECK_HIP(ctx, hipSetDevice(ctx->device));  // the calling thread may be new (one host thread per ctx)
    const sipp_stark_config& cfg = ctx->cfg;
    Shape s;
    SIPP_TRY(shape_of(kind, num_io_in, &s));
    const air_spec_t* a = s.air;
    const FriParamsDev fp = fri_params_of(cfg, s.log_n);
    const uint32_t log_n = s.log_n, log_m = log_n + cfg.rate_bits, R = (uint32_t)fp.arity_bits.size();
    // the layer kernels (transforms, leaf hashing) work on at least 16 values: a limit of this implementation, not of FRI --
    // refused here, before any work, instead of by a kernel wrapper in the middle of the proof
    if (R > 0 && log_m < (R - 1) * cfg.arity_bits + 4)   // the values of the last committed layer
        return sipp_fail(ctx, SIPP_E_UNSUPPORTED, "stark: arity_bits / final_poly_bits fold the FRI layers below 16 values for this trace length");
    const uint32_t log_mq = log_n + 1;                    // the quotient domain: coset 7 <w_2N> = the first 2N leaves of the LDE
    const size_t n = (size_t)1 << log_n, m = (size_t)1 << log_m, mq = (size_t)1 << log_mq;
    const size_t final_len = n >> (R * cfg.arity_bits);
    const int W = s.W, P = s.P, Q = s.Q;
    const uint32_t nq = cfg.num_queries;
    (void)nq;
    const size_t cap_words = (size_t)4 << cfg.cap_height;
    const size_t total_words = proof_words(cfg, s);
    if (proof_cap < total_words) return sipp_fail(ctx, SIPP_E_BUFSZ, "proof buffer too small (see sipp_proof_size)");
    if (n < ((size_t)1 << a->table_bits)) return sipp_fail(ctx, SIPP_E_UNSUPPORTED, "trace shorter than the range table");

    // optional host-side phase timing (SIPP_HOST_TIMING=1): wall clock at each Fiat-Shamir synchronisation point
    static const bool host_timing = getenv("SIPP_HOST_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto tick = [&](const char* what) {
        if (!host_timing) return;
        (void)hipStreamSynchronize(ctx->stream);
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[sipp kind %d] %-18s %8.3f ms\n", kind, what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    ArenaMark mark = arena_mark(ctx);
    struct Release {
        sipp_ctx* c;
        ArenaMark m;
        ~Release() {
            (void)hipStreamSynchronize(c->stream);
            arena_release(c, m);
        }
    } release{ctx, mark};

    uint64_t* pf = proof_out;
    size_t pos = 0;
    auto push = [&](const uint64_t* v, size_t cnt) {
        memcpy(pf + pos, v, cnt * 8);
        pos += cnt;
    };
    {
        uint64_t hdr[16] = {SIPP_MAGIC, (uint64_t)kind, log_n, s.num_io, (uint64_t)W, (uint64_t)P, (uint64_t)Q, cfg.cap_height,
                            R, (uint64_t)final_len, nq, (uint64_t)a->pi_per_io, total_words, cfg.rate_bits, cfg.arity_bits,
                            (uint64_t)(cfg.fs_rule | (cfg.lookup_rule << 1))};
        push(hdr, 16);
    }

    // ---- IO records -> device, trace fill ----
    std::vector<uint32_t> pis;
    uint32_t* d_ios = nullptr;
    SIPP_TRY(upload_ios(ctx, kind, ios, num_io_in, s, &d_ios, &pis));
    int* d_err = arena_alloc_t<int>(ctx, 1);
    uint64_t* d_trace = arena_alloc_t<uint64_t>(ctx, (size_t)W * n);
    Oracle3 T{arena_alloc_t<uint64_t>(ctx, (size_t)W * n), arena_alloc_t<uint64_t>(ctx, (size_t)W * m),
              arena_alloc_t<uint64_t>(ctx, tree_words(log_m)), W};
    Oracle3 Z{arena_alloc_t<uint64_t>(ctx, (size_t)P * n), arena_alloc_t<uint64_t>(ctx, (size_t)P * m),
              arena_alloc_t<uint64_t>(ctx, tree_words(log_m)), P};
    Oracle3 Qo{arena_alloc_t<uint64_t>(ctx, (size_t)2 * mq), arena_alloc_t<uint64_t>(ctx, (size_t)Q * m),
               arena_alloc_t<uint64_t>(ctx, tree_words(log_m)), Q};
    if (!d_err || !d_trace || !T.coeffs || !T.lde || !T.tree || !Z.coeffs || !Z.lde || !Z.tree || !Qo.coeffs || !Qo.lde ||
        !Qo.tree)
        return SIPP_E_NOMEM;
    if (!pis_canonical(kind, pis.data(), s.num_io))
        return sipp_fail(ctx, SIPP_E_WITNESS, "IO record holds a non-canonical field element (>= p)");
    SIPP_CHECK_HIP(ctx, hipMemsetAsync(d_err, 0, sizeof(int), ctx->stream));
    SIPP_TRY(sipp_trace_fill(ctx, a, d_ios, s.num_io, log_n, d_trace, d_err));
    {
        int h_err = 0;
        SIPP_CHECK_HIP(ctx, hipMemcpyAsync(&h_err, d_err, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        SIPP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (h_err) return sipp_fail(ctx, h_err, "trace fill: IO record not provable (claimed output wrong, degenerate point or non-canonical limbs)");
    }

    tick("trace fill");
    if (ctx->gate_release) {
        ctx->gate_release->release();
        ctx->gate_release = nullptr;
    }
    if (ctx->gate_wait) {  // a gated kind without a chain in its trace fill (trace.hip gate_after_chain took the others)
        ctx->gate_wait->wait();
        ctx->gate_wait = nullptr;
    }
    uint64_t cap_host[4 << 8];

    // ---- 1. trace commitment ----
    SIPP_TRY(commit_values_launch(ctx, d_trace, (size_t)W, log_n, T.coeffs, T.lde, T.tree));
    // Fiat-Shamir starts from the statement (hashed on the host while the commitment kernels run)
    // (cfg.fs_rule = SIPP_FS_UPSTREAM: starky's recalled order instead -- the challenger starts at the trace cap)
    host::Challenger ch;
    if (cfg.fs_rule == SIPP_FS_STATEMENT) {
        const uint64_t st[16] = {(uint64_t)kind, log_n, s.num_io, (uint64_t)W, (uint64_t)P, (uint64_t)Q, cfg.rate_bits,
                                 cfg.cap_height, cfg.pow_bits, cfg.arity_bits, cfg.final_poly_bits, cfg.num_queries,
                                 cfg.num_challenges, cfg.pow_rule, (uint64_t)a->pi_per_io, (uint64_t)cfg.lookup_rule};
        uint64_t root[4];
        ch.observe_many(st, 16);
        pi_root(pis.data(), s.num_io, a->pi_per_io, root);
        ch.observe_many(root, 4);
    }
    // public-input polynomials (they depend on the statement alone): interpolated over the order-nio subgroup and shifted HERE, while
    // the host would only wait for the trace commitment's cap (0.4 - 0.65 ms of host time that used to sit between the alphas and
    // the quotient launches of every proof: `profiles/r05_timeline.txt`, the gap in front of ntt_pass_kernel); their LDE to the
    // quotient coset is launched where the quotient needs it
    const int n_aux = a->n_aux;
    const uint32_t log_io = log_n - (uint32_t)a->log_rows;
    const size_t nio = s.num_io;
    std::vector<uint64_t> auxc((size_t)n_aux * nio), col(nio);
    {
        const uint64_t g = gl::root_of_unity(log_n);
        for (int ai = 0; ai < n_aux; ai++) {
            const int word = a->aux[4 * ai], part = a->aux[4 * ai + 1], shift = a->aux[4 * ai + 2], sub = a->aux[4 * ai + 3];
            for (size_t io = 0; io < nio; io++) {
                const uint32_t* rec = &pis[io * a->pi_per_io];
                if (part == 3) {
                    uint16_t limbs[16];
                    fq12_tower_limbs(rec + word, sub / 16, limbs);
                    col[io] = limbs[sub % 16];
                } else {
                    uint32_t w = rec[word];
                    col[io] = part == 0 ? (w & 0xffff) : part == 1 ? (w >> 16) : w;
                }
            }
            host_ifft(col, log_io);
            if (shift) {
                const uint64_t sft = gl::inv(gl::pow(g, (uint64_t)shift));
                uint64_t f = 1;
                for (size_t j = 0; j < nio; j++) {
                    col[j] = gl::mul(col[j], f);
                    f = gl::mul(f, sft);
                }
            }
            memcpy(&auxc[(size_t)ai * nio], col.data(), nio * 8);
        }
    }
    SIPP_TRY(read_cap(ctx, T.tree, log_m, cap_host));
    ch.observe_many(cap_host, cap_words);
    push(cap_host, cap_words);
    tick("trace commit");

    // ---- 2. permutation challenges, Z columns ----
    uint64_t beta[2], gamma[2];
    for (int i = 0; i < 2; i++) {
        beta[i] = ch.get();
        gamma[i] = ch.get();
        if (cfg.lookup_rule == SIPP_LOOKUP_SHARED) beta[i] = gamma[i];     // both factors of a lookup under ONE challenge (sipp_hip.h)
    }
    {
        ArenaMark mz = arena_mark(ctx);
        uint64_t* d_zv = arena_alloc_t<uint64_t>(ctx, (size_t)P * n);
        if (!d_zv) return SIPP_E_NOMEM;
        SIPP_TRY(sipp_k_z_columns(ctx, a, d_trace, log_n, beta, gamma, d_zv));
        SIPP_TRY(commit_values(ctx, d_zv, (size_t)P, log_n, Z.coeffs, Z.lde, Z.tree, cap_host));
        arena_release(ctx, mz);
    }
    ch.observe_many(cap_host, cap_words);
    push(cap_host, cap_words);

    tick("Z + commit");
    // ---- 3. alphas ----
    uint64_t alpha[2];
    alpha[0] = ch.get();
    alpha[1] = ch.get();

    // ---- 4. quotient ----
    {
        ArenaMark mark_q = arena_mark(ctx);
        // (the public-input polynomials' coefficients were interpolated on the host while the trace commitment ran: auxc)
        uint64_t* d_auxc = arena_alloc_t<uint64_t>(ctx, (size_t)n_aux * nio + 1);
        uint64_t* d_aux = arena_alloc_t<uint64_t>(ctx, (size_t)n_aux * mq + 1);
        if (!d_auxc || !d_aux) return SIPP_E_NOMEM;
        SIPP_CHECK_HIP(ctx, hipMemcpyAsync(d_auxc, auxc.data(), auxc.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        SIPP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (n_aux) {
            int rc_aux = sipp_tree_coset_eval(ctx, d_auxc, d_aux, (size_t)n_aux, log_io, log_mq);
            if (rc_aux == SIPP_E_UNSUPPORTED)
                rc_aux = sipp_ntt_dif(ctx, d_auxc, nio, log_io, d_aux, mq, log_mq, (size_t)n_aux, false, NttDiag{gl::GEN, 0});
            SIPP_TRY(rc_aux);
        }
        SIPP_TRY(sipp_k_quotient(ctx, a, log_n, T.lde, Z.lde, m, d_aux, alpha, beta, gamma, Qo.coeffs));
        // coset iNTT: leaf-order values -> natural coefficients of q(x) (undo the shift 7)
        SIPP_TRY(sipp_ntt_dit(ctx, Qo.coeffs, mq, log_mq, 2, true, NttDiag{gl::inv(gl::GEN), 0}));
        // [2][2N] natural == 4 chunks of N coefficients, contiguous
        SIPP_TRY(commit_coeffs(ctx, Qo.coeffs, (size_t)Q, log_n, Qo.lde, Qo.tree, cap_host));
        arena_release(ctx, mark_q);
    }
    ch.observe_many(cap_host, cap_words);
    push(cap_host, cap_words);

    tick("quotient + commit");
    // ---- 5. zeta, openings ----
    const gl::E2 zeta = ch.get_ext();
    if (gl::eq(gl::pow(zeta, (uint64_t)n), gl::e2(1))) return sipp_fail(ctx, SIPP_E_SUBGROUP, "zeta in the trace subgroup");
    const gl::E2 gzeta = gl::scale(zeta, gl::root_of_unity(log_n));
    uint64_t* d_zp[2] = {arena_alloc_t<uint64_t>(ctx, 2 * n), arena_alloc_t<uint64_t>(ctx, 2 * n)};
    uint64_t* d_zip[2] = {arena_alloc_t<uint64_t>(ctx, 2 * n), arena_alloc_t<uint64_t>(ctx, 2 * n)};
    uint64_t* d_open = arena_alloc_t<uint64_t>(ctx, (size_t)(W + P + Q) * 4);
    if (!d_zp[0] || !d_zp[1] || !d_zip[0] || !d_zip[1] || !d_open) return SIPP_E_NOMEM;
    {
        const gl::E2 bases[4] = {zeta, gzeta, gl::inv(zeta), gl::inv(gzeta)};
        uint64_t* const tabs[4] = {d_zp[0], d_zp[1], d_zip[0], d_zip[1]};
        SIPP_TRY(sipp_k_pow_table4(ctx, bases, n, tabs));
    }
    {
        const uint64_t* const cf[3] = {T.coeffs, Z.coeffs, Qo.coeffs};
        const uint32_t nc3[3] = {(uint32_t)W, (uint32_t)P, (uint32_t)Q};
        SIPP_TRY(sipp_k_openings3(ctx, cf, nc3, n, d_zp[0], d_zp[1], d_open));
    }
    std::vector<uint64_t> hop((size_t)(W + P + Q) * 4);
    SIPP_CHECK_HIP(ctx, hipMemcpyAsync(hop.data(), d_open, hop.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    SIPP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    {
        // proof order: local[W] next[W] z[P] z_next[P] quotient[Q]
        for (int c = 0; c < W; c++) push(&hop[(size_t)c * 4], 2);
        for (int c = 0; c < W; c++) push(&hop[(size_t)c * 4 + 2], 2);
        for (int c = 0; c < P; c++) push(&hop[(size_t)(W + c) * 4], 2);
        for (int c = 0; c < P; c++) push(&hop[(size_t)(W + c) * 4 + 2], 2);
        for (int c = 0; c < Q; c++) push(&hop[(size_t)(W + P + c) * 4], 2);
        // observe: batch 0 = local | z | quotient ; batch 1 = next | z_next
        for (int c = 0; c < W; c++) ch.observe_many(&hop[(size_t)c * 4], 2);
        for (int c = 0; c < P; c++) ch.observe_many(&hop[(size_t)(W + c) * 4], 2);
        for (int c = 0; c < Q; c++) ch.observe_many(&hop[(size_t)(W + P + c) * 4], 2);
        for (int c = 0; c < W; c++) ch.observe_many(&hop[(size_t)c * 4 + 2], 2);
        for (int c = 0; c < P; c++) ch.observe_many(&hop[(size_t)(W + c) * 4 + 2], 2);
    }

    tick("openings+observe");
    // ---- 6. FRI ----
    const gl::E2 fa = ch.get_ext();
    uint64_t* d_final = arena_alloc_t<uint64_t>(ctx, 2 * n);
    {
        // limbs of alpha^c (c0, c1) for the lazy combination (prover.hip::fri_combine_kernel)
        std::vector<uint32_t> apow((size_t)(W + P + Q) * 6);
        gl::E2 ap = gl::e2(1);
        for (int c = 0; c < W + P + Q; c++) {
            const uint64_t comp[2] = {ap.c0, ap.c1};
            for (int q = 0; q < 2; q++) {
                apow[6 * c + 3 * q] = (uint32_t)comp[q] & 0x3FFFFFu;
                apow[6 * c + 3 * q + 1] = (uint32_t)(comp[q] >> 22) & 0x3FFFFFu;
                apow[6 * c + 3 * q + 2] = (uint32_t)(comp[q] >> 44);
            }
            ap = gl::mul(ap, fa);
        }
        uint32_t* d_apow = arena_alloc_t<uint32_t>(ctx, apow.size());
        if (!d_final || !d_apow) return SIPP_E_NOMEM;
        SIPP_CHECK_HIP(ctx, hipMemcpyAsync(d_apow, apow.data(), apow.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        SIPP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const uint64_t* src[3] = {T.coeffs, Z.coeffs, Qo.coeffs};
        const int cnt[3] = {W, P, Q};
        const uint64_t* zp[2] = {d_zp[0], d_zp[1]};
        const uint64_t* zip[2] = {d_zip[0], d_zip[1]};
        SIPP_TRY(sipp_k_fri_final(ctx, src, cnt, n, d_apow, W + P, gl::pow(fa, (uint64_t)(W + P)), zp, zip, d_final));
    }
    // commit phase, proof of work, query rounds: the generic FRI core (fri.hip)
    {
        const FriOracleDev ors[3] = {{T.lde, m, (uint32_t)W, T.tree}, {Z.lde, m, (uint32_t)P, Z.tree}, {Qo.lde, m, (uint32_t)Q, Qo.tree}};
        size_t flen = 0;
        SIPP_TRY(sipp_fri_prove_core(ctx, ors, 3, log_n, fp, d_final, ch, pf, pos, total_words, &flen, tick));
        if (flen != final_len) return sipp_fail(ctx, SIPP_E_BUFSZ, "internal: final polynomial length mismatch");
    }
    for (size_t k = 0; k < pis.size(); k++) pf[pos++] = pis[k];
    tick("queries+assemble");
    if (pos != total_words) return sipp_fail(ctx, SIPP_E_BUFSZ, "internal: proof length mismatch");
    *proof_len = pos;
    return SIPP_OK;
}

extern "C" {

int sipp_stark_shape(const sipp_ctx* ctx, int kind, size_t num_io, uint32_t* log_rows, uint32_t* main_cols,
                     uint32_t* perm_cols, uint32_t* quotient_cols) {
    kind = ctx_kind(ctx, kind);
    Shape s;
    SIPP_TRY(shape_of(kind, num_io, &s));
    if (log_rows) *log_rows = s.log_n;
    if (main_cols) *main_cols = (uint32_t)s.W;
    if (perm_cols) *perm_cols = (uint32_t)s.P;
    if (quotient_cols) *quotient_cols = (uint32_t)s.Q;
    return SIPP_OK;
}

int sipp_io_shard(size_t num_io, uint32_t world, uint32_t rank, size_t* first, size_t* count) {
    if (!first || !count || world == 0 || rank >= world || num_io > SIZE_MAX / world) return SIPP_E_BADARG;
    // balanced contiguous ranges: floor(rank n / world) .. floor((rank + 1) n / world); the ranges tile [0, n) exactly
    const size_t lo = num_io * rank / world, hi = num_io * ((size_t)rank + 1) / world;
    *first = lo;
    *count = hi - lo;
    return SIPP_OK;
}

size_t sipp_workspace_bytes(int kind, size_t num_io) { return sipp_workspace_bytes_cfg(kind, num_io, nullptr); }

int sipp_device_memory(int device, size_t* free_bytes, size_t* total_bytes) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return SIPP_E_HIP;
    int prev = 0;
    (void)hipGetDevice(&prev);
    size_t fr = 0, tot = 0;
    if (hipSetDevice(device) != hipSuccess || hipMemGetInfo(&fr, &tot) != hipSuccess) return SIPP_E_HIP;
    (void)hipSetDevice(prev);
    if (free_bytes) *free_bytes = fr;
    if (total_bytes) *total_bytes = tot;
    return SIPP_OK;
}

size_t sipp_workspace_bytes_cfg(int kind, size_t num_io, const sipp_stark_config* cfg) {
    Shape s;
    if (shape_of(kind, num_io, &s) != SIPP_OK) return 0;
    const uint32_t rate_bits = cfg ? cfg->rate_bits : 1;
    if (rate_bits < 1 || rate_bits > 3) return 0;
    const size_t n = (size_t)1 << s.log_n, m = n << rate_bits;
    const size_t W = (size_t)s.W, P = (size_t)s.P, Q = (size_t)s.Q, nc = (size_t)s.air->n_checked;
    size_t words = n * (2 * W + 3 * P)                                  // trace values + coefficients, Z values (x2) + coefficients
                   + m * (W + P + Q + 2 + (size_t)s.air->n_aux)         // LDEs, quotient values, public-input LDEs
                   + 3 * 8 * m                                          // three Merkle trees
                   + 2 * m * (1 + (size_t)s.air->n_gadgets + (size_t)s.air->n_constraints / 64 + 2 +
                              (kind == SIPP_PAIRING ? SIPP_QUOTIENT_MAX_SLICES : 0))  // quotient segment partials (+ the slices of a many-product gadget)
                   + (sipp_quotient_rest_chunks(s.log_n, s.air->n_checked) > 1                 // thin QUOTIENT domains (2N points, whatever
                          ? (size_t)sipp_quotient_rest_chunks(s.log_n, s.air->n_checked) * 6 * 2 * n : 0)   // the blowup): ranges x 6 sums
                   + (W + P + Q) * SIPP_OPENINGS_MAX_SEGS * 4               // partial sums of the grouped openings
                   + 80 * n;                                            // power tables, FRI layers, combine partials
    size_t bytes = 8 * words + nc * ((size_t)16 << s.air->table_bits)   // lookup histogram / scan scratch
                   + n * 400                                            // Jacobian row scratch of the curve chains
                   + ((size_t)64 << 20);
    return bytes + bytes / 16;
}

size_t sipp_proof_size(const sipp_ctx* ctx, int kind, size_t num_io) {
    kind = ctx_kind(ctx, kind);
    Shape s;
    if (!ctx || shape_of(kind, num_io, &s) != SIPP_OK) return 0;
    return proof_words(ctx->cfg, s);
}

int sipp_trace_build(sipp_ctx* ctx, int kind, const uint32_t* ios, size_t num_io, uint64_t* d_trace) {
    if (!ctx || !ios || !d_trace) return SIPP_E_BADARG;
    kind = ctx_kind(ctx, kind);
    SIPP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    Shape s;
    SIPP_TRY(shape_of(kind, num_io, &s));
    ArenaMark m = arena_mark(ctx);
    uint32_t* d_ios = nullptr;
    int rc = upload_ios(ctx, kind, ios, num_io, s, &d_ios, nullptr);
    int* d_err = arena_alloc_t<int>(ctx, 1);
    if (rc == SIPP_OK && !d_err) rc = SIPP_E_NOMEM;
    if (rc == SIPP_OK) {
        SIPP_CHECK_HIP(ctx, hipMemsetAsync(d_err, 0, sizeof(int), ctx->stream));
        rc = sipp_trace_fill(ctx, s.air, d_ios, s.num_io, s.log_n, d_trace, d_err);
    }
    if (rc == SIPP_OK) {
        int h_err = 0;
        SIPP_CHECK_HIP(ctx, hipMemcpyAsync(&h_err, d_err, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        rc = sipp_sync(ctx);
        if (rc == SIPP_OK && h_err) rc = sipp_fail(ctx, h_err, "trace fill: IO record not provable");
    }
    arena_release(ctx, m);
    return rc;
}

int sipp_exp_outputs(sipp_ctx* ctx, int kind, uint32_t* ios, size_t num_io) {
    if (!ctx || !ios) return SIPP_E_BADARG;
    SIPP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    kind = base_kind(kind);   // the outputs of the hardened kinds are the plain chains'
    Shape s;
    SIPP_TRY(shape_of(kind, num_io, &s));
    const size_t ppi = IO_WORDS[kind], out_words = kind == SIPP_G1_EXP ? 16 : (kind == SIPP_FQ12_EXP || kind == SIPP_PAIRING) ? 96 : 32;
    ArenaScope scope(ctx);
    uint32_t* d_ios = nullptr;
    int rc = upload_ios(ctx, kind, ios, num_io, s, &d_ios, nullptr);
    int* d_err = arena_alloc_t<int>(ctx, 1);
    // the curves' outputs come straight from the accumulator chains (no trace rows); the Fq12 chain writes its cells
    uint64_t* d_trace = kind == SIPP_FQ12_EXP ? arena_alloc_t<uint64_t>(ctx, (size_t)s.W << s.log_n) : nullptr;
    if (rc == SIPP_OK && (!d_err || (kind == SIPP_FQ12_EXP && !d_trace))) rc = SIPP_E_NOMEM;
    if (rc == SIPP_OK) {
        SIPP_CHECK_HIP(ctx, hipMemsetAsync(d_err, 0, sizeof(int), ctx->stream));
        ctx->outputs_only = true;
        rc = sipp_trace_fill(ctx, s.air, d_ios, s.num_io, s.log_n, d_trace, d_err);
        ctx->outputs_only = false;
    }
    if (rc == SIPP_OK) {
        int h_err = 0;
        uint32_t* h = reinterpret_cast<uint32_t*>(ctx->h_pinned);
        SIPP_CHECK_HIP(ctx, hipMemcpyAsync(h, d_ios, num_io * ppi * 4, hipMemcpyDeviceToHost, ctx->stream));
        SIPP_CHECK_HIP(ctx, hipMemcpyAsync(&h_err, d_err, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        rc = sipp_sync(ctx);
        if (rc == SIPP_OK && h_err) rc = sipp_fail(ctx, h_err, "exp_outputs: IO record not provable");
        if (rc == SIPP_OK)
            for (size_t io = 0; io < num_io; io++)
                memcpy(ios + io * ppi + (ppi - out_words), h + io * ppi + (ppi - out_words), out_words * 4);
    }
    return rc;
}

}  // extern "C"

// A' = A1 + [x] A2 and B' = B1 + [1/x] B2 of one SIPP round: both are latency-bound chains of 255 doublings on a few
// workgroups, so the G1 chain runs on a second stream beside the G2 chain.
int sipp_fold_outputs(sipp_ctx* ctx, uint32_t* g1_ios, size_t n1, uint32_t* g2_ios, size_t n2) {
    if (!ctx || !g1_ios || !g2_ios) return SIPP_E_BADARG;
    SIPP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->aux_stream) SIPP_CHECK_HIP(ctx, hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking));
    Shape s[2];
    SIPP_TRY(shape_of(SIPP_G1_EXP, n1, &s[0]));
    SIPP_TRY(shape_of(SIPP_G2_EXP, n2, &s[1]));
    uint32_t* ios[2] = {g1_ios, g2_ios};
    const size_t num[2] = {n1, n2}, ppi[2] = {SIPP_G1_IO_WORDS, SIPP_G2_IO_WORDS}, outw[2] = {16, 32};
    // each list is read back into its own HALF of the pinned buffer (h_pinned_words u32 each); the last two u32 of the
    // buffer hold the error words
    for (int k = 0; k < 2; k++)
        if (num[k] * ppi[k] + 2 > ctx->h_pinned_words)
            return sipp_fail(ctx, SIPP_E_NOMEM, "fold_outputs: obligation list larger than half of the pinned staging buffer");
    ArenaScope scope(ctx);
    uint32_t* d_ios[2] = {nullptr, nullptr};
    int* d_err = arena_alloc_t<int>(ctx, 2);
    if (!d_err) return SIPP_E_NOMEM;
    int rc = SIPP_OK;
    for (int k = 0; k < 2 && rc == SIPP_OK; k++) rc = upload_ios(ctx, k, ios[k], num[k], s[k], &d_ios[k], nullptr);
    hipStream_t main_stream = ctx->stream;
    uint32_t* h = reinterpret_cast<uint32_t*>(ctx->h_pinned);
    const size_t h_off[2] = {0, ctx->h_pinned_words};  // u32 offsets: the two halves of the pinned buffer
    // the error words land in pinned memory too: a device-to-host copy into pageable memory would block the host until the
    // first chain is done, i.e. before the second one is even launched
    volatile int* h_err = reinterpret_cast<volatile int*>(h + 2 * ctx->h_pinned_words - 2);
    h_err[0] = h_err[1] = 0;
    if (rc == SIPP_OK) {
        SIPP_CHECK_HIP(ctx, hipMemsetAsync(d_err, 0, 2 * sizeof(int), main_stream));
        SIPP_CHECK_HIP(ctx, hipStreamSynchronize(main_stream));
        ctx->outputs_only = true;
        for (int k = 0; k < 2 && rc == SIPP_OK; k++) {
            ctx->stream = k == 0 ? ctx->aux_stream : main_stream;
            rc = sipp_trace_fill(ctx, s[k].air, d_ios[k], s[k].num_io, s[k].log_n, nullptr, d_err + k);
            // sipp_trace_fill has handed its row scratch back, but on another stream the kernels are still using it:
            // take exactly that block again so that the next chain gets its own
            if (rc == SIPP_OK && !arena_alloc(ctx, sipp_curve_rows_bytes(k, s[k].log_n))) rc = SIPP_E_NOMEM;
            if (rc == SIPP_OK) {
                (void)hipMemcpyAsync(h + h_off[k], d_ios[k], num[k] * ppi[k] * 4, hipMemcpyDeviceToHost, ctx->stream);
                (void)hipMemcpyAsync(const_cast<int*>(h_err) + k, d_err + k, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
            }
        }
        ctx->outputs_only = false;
        ctx->stream = main_stream;
        const hipError_t e0 = hipStreamSynchronize(ctx->aux_stream), e1 = hipStreamSynchronize(main_stream);
        if (rc == SIPP_OK && (e0 != hipSuccess || e1 != hipSuccess)) rc = sipp_fail(ctx, SIPP_E_HIP, "fold_outputs: stream synchronisation failed");
    }
    if (rc == SIPP_OK && (h_err[0] || h_err[1])) rc = sipp_fail(ctx, h_err[0] ? h_err[0] : h_err[1], "fold_outputs: obligation not provable");
    if (rc == SIPP_OK)
        for (int k = 0; k < 2; k++)
            for (size_t io = 0; io < num[k]; io++)
                memcpy(ios[k] + io * ppi[k] + (ppi[k] - outw[k]), h + h_off[k] + io * ppi[k] + (ppi[k] - outw[k]), outw[k] * 4);
    return rc;
}

// ---- two-phase folds (see ctx.hpp) ----
static int fold_upload(sipp_ctx* ctx, int k, const uint32_t* ios, size_t num, uint32_t padded, uint32_t* d_dst, hipStream_t st) {
    const size_t ppi = IO_WORDS[k], words = (size_t)padded * ppi;
    uint32_t* h = reinterpret_cast<uint32_t*>(ctx->h_pinned) + (k ? ctx->h_pinned_words : 0);   // each list has its own half
    for (size_t io = 0; io < padded; io++) memcpy(h + io * ppi, ios + (io < num ? io : num - 1) * ppi, ppi * 4);
    SIPP_CHECK_HIP(ctx, hipMemcpyAsync(d_dst, h, words * 4, hipMemcpyHostToDevice, st));
    return SIPP_OK;
}

int sipp_fold_begin(sipp_ctx* ctx, const uint32_t* g1_ios, size_t n1, const uint32_t* g2_ios, size_t n2) {
    if (!ctx || !g1_ios || !g2_ios) return SIPP_E_BADARG;
    if (ctx->fold.active) return sipp_fail(ctx, SIPP_E_BADARG, "fold_begin: a fold is already in flight");
    SIPP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->aux_stream) SIPP_CHECK_HIP(ctx, hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking));
    if (!ctx->aux2_stream) SIPP_CHECK_HIP(ctx, hipStreamCreateWithFlags(&ctx->aux2_stream, hipStreamNonBlocking));
    Shape s[2];
    SIPP_TRY(shape_of(SIPP_G1_EXP, n1, &s[0]));
    SIPP_TRY(shape_of(SIPP_G2_EXP, n2, &s[1]));
    const uint32_t* ios[2] = {g1_ios, g2_ios};
    const size_t num[2] = {n1, n2};
    for (int k = 0; k < 2; k++)
        if ((size_t)s[k].num_io * IO_WORDS[k] + 2 > ctx->h_pinned_words)
            return sipp_fail(ctx, SIPP_E_NOMEM, "fold: obligation list larger than half of the pinned staging buffer");
    sipp_ctx::Fold& f = ctx->fold;
    f.mark = ctx->arena_off;
    f.d_err = arena_alloc_t<int>(ctx, 2);
    for (int k = 0; k < 2; k++) {
        f.num_io[k] = s[k].num_io;
        f.n_in[k] = num[k];
        f.d_ios[k] = arena_alloc_t<uint32_t>(ctx, (size_t)s[k].num_io * IO_WORDS[k]);
        f.rows[k] = arena_alloc(ctx, sipp_fold_rows_bytes(k, s[k].num_io));
    }
    if (!f.d_err || !f.d_ios[0] || !f.d_ios[1] || !f.rows[0] || !f.rows[1]) {
        ctx->arena_off = f.mark;
        return SIPP_E_NOMEM;
    }
    hipStream_t st[2] = {ctx->aux_stream, ctx->aux2_stream};
    int rc = SIPP_OK;
    // earlier work on the main stream may still be using the arena block handed out here: order the side streams behind it
    (void)hipStreamSynchronize(ctx->stream);
    for (int k = 0; k < 2 && rc == SIPP_OK; k++) {
        rc = fold_upload(ctx, k, ios[k], num[k], s[k].num_io, f.d_ios[k], st[k]);
        // the staging buffer belongs to the caller again when begin returns (sipp_exp_outputs and the provers stage through it)
        if (rc == SIPP_OK && hipStreamSynchronize(st[k]) != hipSuccess) rc = sipp_fail(ctx, SIPP_E_HIP, "fold_begin: upload failed");
        if (rc == SIPP_OK) rc = sipp_fold_chain_begin(ctx, k, f.d_ios[k], s[k].num_io, IO_WORDS[k], f.rows[k], st[k]);
    }
    if (rc != SIPP_OK) {
        (void)hipStreamSynchronize(st[0]);
        (void)hipStreamSynchronize(st[1]);
        ctx->arena_off = f.mark;
        return rc;
    }
    f.active = true;
    return SIPP_OK;
}

int sipp_fold_finish(sipp_ctx* ctx, uint32_t* g1_ios, size_t n1, uint32_t* g2_ios, size_t n2) {
    if (!ctx || !g1_ios || !g2_ios) return SIPP_E_BADARG;
    sipp_ctx::Fold& f = ctx->fold;
    if (!f.active || f.n_in[0] != n1 || f.n_in[1] != n2) return sipp_fail(ctx, SIPP_E_BADARG, "fold_finish: no matching fold_begin");
    SIPP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st[2] = {ctx->aux_stream, ctx->aux2_stream};
    uint32_t* ios[2] = {g1_ios, g2_ios};
    const size_t num[2] = {n1, n2}, outw[2] = {16, 32};
    uint32_t* h = reinterpret_cast<uint32_t*>(ctx->h_pinned);
    volatile int* h_err = reinterpret_cast<volatile int*>(h + 2 * ctx->h_pinned_words - 2);
    int rc = SIPP_OK;
    // the staging halves are free again only when the uploads of begin have completed: the chains are long done by now
    for (int k = 0; k < 2; k++) (void)hipStreamSynchronize(st[k]);
    h_err[0] = h_err[1] = 0;
    for (int k = 0; k < 2 && rc == SIPP_OK; k++) {
        const size_t ppi = IO_WORDS[k];
        (void)hipMemsetAsync(f.d_err + k, 0, sizeof(int), st[k]);
        rc = fold_upload(ctx, k, ios[k], num[k], f.num_io[k], f.d_ios[k], st[k]);     // the complete records (exponents now known)
        if (rc == SIPP_OK) rc = sipp_fold_chain_finish(ctx, k, f.d_ios[k], f.num_io[k], (uint32_t)ppi, f.rows[k], f.d_err + k, st[k]);
    }
    for (int k = 0; k < 2; k++) (void)hipStreamSynchronize(st[k]);   // uploads read the staging halves the downloads write
    if (rc == SIPP_OK)
        for (int k = 0; k < 2; k++) {
            const size_t ppi = IO_WORDS[k];
            (void)hipMemcpyAsync(h + (k ? ctx->h_pinned_words : 0), f.d_ios[k], num[k] * ppi * 4, hipMemcpyDeviceToHost, st[k]);
            (void)hipMemcpyAsync(const_cast<int*>(h_err) + k, f.d_err + k, sizeof(int), hipMemcpyDeviceToHost, st[k]);
        }
    const hipError_t e0 = hipStreamSynchronize(st[0]), e1 = hipStreamSynchronize(st[1]);
    if (rc == SIPP_OK && (e0 != hipSuccess || e1 != hipSuccess)) rc = sipp_fail(ctx, SIPP_E_HIP, "fold_finish: stream synchronisation failed");
    if (rc == SIPP_OK && (h_err[0] || h_err[1])) rc = sipp_fail(ctx, h_err[0] ? h_err[0] : h_err[1], "fold: obligation not provable");
    if (rc == SIPP_OK)
        for (int k = 0; k < 2; k++) {
            const size_t ppi = IO_WORDS[k];
            const uint32_t* src = h + (k ? ctx->h_pinned_words : 0);
            for (size_t io = 0; io < num[k]; io++)
                memcpy(ios[k] + io * ppi + (ppi - outw[k]), src + io * ppi + (ppi - outw[k]), outw[k] * 4);
        }
    ctx->arena_off = f.mark;
    f.active = false;
    return rc;
}

extern "C" {

int sipp_g1_exp_prove(sipp_ctx* ctx, const uint32_t* ios, size_t num_io, uint64_t* proof_out, size_t proof_cap,
                      size_t* proof_len) {
    return prove_impl(ctx, SIPP_G1_EXP, ios, num_io, proof_out, proof_cap, proof_len);
}
int sipp_g2_exp_prove(sipp_ctx* ctx, const uint32_t* ios, size_t num_io, uint64_t* proof_out, size_t proof_cap,
                      size_t* proof_len) {
    return prove_impl(ctx, SIPP_G2_EXP, ios, num_io, proof_out, proof_cap, proof_len);
}
int sipp_fq12_exp_prove(sipp_ctx* ctx, const uint32_t* ios, size_t num_io, uint64_t* proof_out, size_t proof_cap,
                        size_t* proof_len) {
    return prove_impl(ctx, SIPP_FQ12_EXP, ios, num_io, proof_out, proof_cap, proof_len);
}
// any kind, synchronously (the hardened G1 / G2 kinds have no entry point of their own)
int sipp_prove(sipp_ctx* ctx, int kind, const uint32_t* ios, size_t num_io, uint64_t* proof_out, size_t proof_cap, size_t* proof_len) {
    if (kind < SIPP_G1_EXP || kind > SIPP_PAIRING) return ctx ? sipp_fail(ctx, SIPP_E_BADARG, "sipp_prove: unknown kind") : SIPP_E_BADARG;
    return prove_impl(ctx, kind, ios, num_io, proof_out, proof_cap, proof_len);
}

// MapToG2 (kind 3, include/sipp_hip.h): the STARK behind batch_map_to_g2_circuit (reference src/bin/bls_aggregation.rs:65)
int sipp_map_to_g2_prove(sipp_ctx* ctx, const uint32_t* ios, size_t num_io, uint64_t* proof_out, size_t proof_cap,
                         size_t* proof_len) {
    return prove_impl(ctx, SIPP_MAP_G2, ios, num_io, proof_out, proof_cap, proof_len);
}
int sipp_pairing_prove(sipp_ctx* ctx, const uint32_t* ios, size_t num_io, uint64_t* proof_out, size_t proof_cap, size_t* proof_len) {
    return prove_impl(ctx, SIPP_PAIRING, ios, num_io, proof_out, proof_cap, proof_len);
}

// ---- asynchronous form -------------------------------------------------------------------------------------
// plonky2 runs its witness generators serially on one thread (SURVEY.md section 8b "who calls it"): a patched caller
// starts the three proofs on three ctxs as soon as the IO values exist and collects them afterwards.  The prover
// needs a host thread of its own (Fiat-Shamir challenges are derived on the host between phases), so every ctx
// owns one worker, created on the first sipp_prove_async and joined by sipp_ctx_destroy.
static void async_worker(sipp_ctx* ctx) {
    sipp_ctx::Async& a = ctx->async;
    std::unique_lock<std::mutex> lk(a.mu);
    for (;;) {
        a.cv.wait(lk, [&] { return a.quit || (a.has_job && !a.done); });
        if (a.quit) return;
        lk.unlock();
        size_t len = 0;
        const int rc = prove_impl(ctx, a.kind, a.ios, a.num_io, a.out, a.cap, &len);
        lk.lock();
        a.rc = rc;
        a.len = len;
        a.done = true;
        a.cv.notify_all();
    }
}

int sipp_prove_async(sipp_ctx* ctx, int kind, const uint32_t* ios, size_t num_io, uint64_t* proof_out, size_t proof_cap) {
    if (!ctx || !ios || !proof_out) return SIPP_E_BADARG;
    if (kind < SIPP_G1_EXP || kind > SIPP_PAIRING) return sipp_fail(ctx, SIPP_E_BADARG, "prove_async: unknown kind");
    sipp_ctx::Async& a = ctx->async;
    std::unique_lock<std::mutex> lk(a.mu);
    if (a.has_job) {
        lk.unlock();
        return sipp_fail(ctx, SIPP_E_BADARG, "prove_async: a proof is already in flight on this ctx (call sipp_wait first)");
    }
    a.kind = kind;
    a.ios = ios;
    a.num_io = num_io;
    a.out = proof_out;
    a.cap = proof_cap;
    a.len = 0;
    a.rc = SIPP_OK;
    a.done = false;
    a.has_job = true;
    if (!a.started) {
        a.started = true;
        a.th = std::thread(async_worker, ctx);
    }
    a.cv.notify_all();
    return SIPP_OK;
}

int sipp_wait(sipp_ctx* ctx, size_t* proof_len) {
    if (!ctx) return SIPP_E_BADARG;
    sipp_ctx::Async& a = ctx->async;
    std::unique_lock<std::mutex> lk(a.mu);
    if (!a.has_job) {
        lk.unlock();
        return sipp_fail(ctx, SIPP_E_BADARG, "wait: no proof in flight on this ctx");
    }
    a.cv.wait(lk, [&] { return a.done; });
    a.has_job = false;
    if (proof_len) *proof_len = a.len;
    return a.rc;
}

int sipp_instance_prove(sipp_ctx* const ctxs[3], const uint32_t* const ios[3], const size_t num_io[3],
                        uint64_t* const proof_out[3], const size_t proof_cap[3], size_t proof_len[3]) {
    if (!ctxs || !ios || !num_io || !proof_out || !proof_cap || !proof_len) return SIPP_E_BADARG;
    for (int k = 0; k < 3; k++)
        if (!ctxs[k]) return SIPP_E_BADARG;
    if (ctxs[0] == ctxs[1] && ctxs[1] == ctxs[2]) {
        // ONE ctx for the instance: the three proofs back to back on its stream and its arena (sized for the largest of them) -- for
        // instances whose three arenas do not fit the card together (n = 4096 with the hardened AIRs: 276 GB of 288; back to back
        // 179 GB).  Three streams are worth 5.5 % at n = 4096 and 13 - 15 % at n = 1024 (scripts/large_n_modes.py), so this is the
        // fallback, not the default.  Largest first, like the concurrent order.
        const int serial[3] = {SIPP_G2_EXP, SIPP_G1_EXP, SIPP_FQ12_EXP};
        for (int k = 0; k < 3; k++) proof_len[k] = 0;
        int first_rc = SIPP_OK;      // like the three-ctx path: every proof is attempted, the first failing status is returned
        for (int i = 0; i < 3; i++) {
            const int k = serial[i];
            if (num_io[k] == 0) continue;
            const int rc = sipp_prove(ctxs[k], k, ios[k], num_io[k], proof_out[k], proof_cap[k], &proof_len[k]);
            if (rc != SIPP_OK) {
                proof_len[k] = 0;
                if (first_rc == SIPP_OK) first_rc = rc;
            }
        }
        return first_rc;
    }
    if (ctxs[0] == ctxs[1] || ctxs[0] == ctxs[2] || ctxs[1] == ctxs[2]) return SIPP_E_BADARG;
    // The other two proofs start once the longest (G2) has its trace filled: its latency-bound chains and lookup kernels
    // are otherwise crowded out by the others' long-running hash workgroups and G2 ends last by ~10 ms (n = 128: 73.3 ->
    // 70.2 ms per instance; n = 1024: neutral; releasing after G2's trace COMMIT instead: 75-81 ms; Fq12 first: 71-72 ms).
    const int first = SIPP_G2_EXP;
    // round 2 (fused LDE kernels): only G1 waits -- 69.0-69.6 ms per n = 128 instance against 70.0-70.6 with Fq12 gated too
    // (re-swept in round 3 on the new kernels: no gate 60.6-61.4, Fq12 gated too 69.2-69.5 against 58.2-58.9 ms)
    // (round 4, after the thin trees' hasher moved to the matrix pipe, scripts/ab_prebuilt.sh, three alternating passes: no gate
    // 57.9-58.9 against 56.4-56.6 ms single, the queue of five 49.3-49.6 either way)
    constexpr int gate_mask = 1 << SIPP_G1_EXP;
    const int order[3] = {SIPP_G2_EXP, SIPP_FQ12_EXP, SIPP_G1_EXP};  // the largest proof first
    sipp_gate gate;
    int started[3] = {0, 0, 0}, rc = SIPP_OK;
    for (int k = 0; k < 3; k++) {
        ctxs[k]->gate_release = (k == first && gate_mask) ? &gate : nullptr;
        ctxs[k]->gate_wait = (gate_mask >> k & 1) ? &gate : nullptr;
    }
    for (int k = 0; k < 3; k++) proof_len[k] = 0;
    for (int i = 0; i < 3 && rc == SIPP_OK; i++) {
        const int k = order[i];
        if (num_io[k] == 0) continue;  // an IO shard (sipp_io_shard) may hold no record of a kind: no proof of that kind
        rc = sipp_prove_async(ctxs[k], k, ios[k], num_io[k], proof_out[k], proof_cap[k]);
        started[k] = rc == SIPP_OK;
    }
    if (!started[first]) gate.release();  // nobody is left to open it
    for (int k = 0; k < 3; k++) {
        if (!started[k]) continue;
        const int r = sipp_wait(ctxs[k], &proof_len[k]);  // every started proof is collected, also after a failure
        if (rc == SIPP_OK) rc = r;
    }
    for (int k = 0; k < 3; k++) ctxs[k]->gate_wait = ctxs[k]->gate_release = nullptr;
    return rc;
}

int sipp_instances_prove(sipp_ctx* const* ctxs, size_t in_flight, size_t count, const uint32_t* const* ios, const size_t* num_io,
                         uint64_t* const* proof_out, const size_t* proof_cap, size_t* proof_len, int* status) {
    if (!ctxs || !in_flight || !ios || !num_io || !proof_out || !proof_cap || !proof_len) return SIPP_E_BADARG;
    for (size_t i = 0; i < 3 * count; i++)
        if (num_io[i] && (!ios[i] || !proof_out[i])) return SIPP_E_BADARG;   // a kind without records needs no buffers
    for (size_t i = 0; i < 3 * in_flight; i++) {
        if (!ctxs[i]) return SIPP_E_BADARG;
        for (size_t j = 0; j < i; j++)
            if (ctxs[i] == ctxs[j]) return SIPP_E_BADARG;
    }
    // one host thread per slot of three ctxs; the slots take the instances from a shared counter, so that the latency-bound
    // head and tail of one instance overlap the hashing of the others (n = 128 on one MI355X: 57 ms per instance with three
    // slots against 62 ms one at a time)
    std::atomic<size_t> next{0};
    std::atomic<int> first_rc{SIPP_OK};
    auto worker = [&](size_t slot) {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= count) return;
            const int rc = sipp_instance_prove(ctxs + 3 * slot, ios + 3 * i, num_io + 3 * i, proof_out + 3 * i, proof_cap + 3 * i,
                                               proof_len + 3 * i);
            if (status) status[i] = rc;
            int ok = SIPP_OK;
            if (rc != SIPP_OK) first_rc.compare_exchange_strong(ok, rc);
        }
    };
    // no exception may cross the C ABI, and a joinable std::thread must not be destroyed: if a thread cannot be created (or the
    // vector cannot grow), the threads that did start plus this one drain the queue -- fewer slots, same result
    std::vector<std::thread> pool;
    const size_t slots = in_flight < count ? in_flight : count;
    try {
        pool.reserve(slots);
        for (size_t sl = 1; sl < slots; sl++) pool.emplace_back(worker, sl);
    } catch (...) {
    }
    if (slots) worker(0);
    for (auto& t : pool) t.join();
    return first_rc.load();
}
}
