// sipp_amd/csrc/ctx.hpp -- internal state behind the opaque sipp_ctx of include/sipp_hip.h.
//
// One ctx == one GPU == one HIP stream.  All device memory comes from a bump arena that is
// reserved once at sipp_ctx_create (no hipMalloc in the steady state, SURVEY.md section 8b
// "threading" row); twiddle / power tables are cached per (kind, size) for the ctx lifetime.
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <condition_variable>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "../../include/sipp_hip.h"
#include "gl.hpp"

// one-shot host-side gate between the proofs of one SIPP instance (sipp_instance_prove): the shortest proof starts once
// the longest has its trace filled, so that the long proof's latency-bound first phase is not crowded out
struct sipp_gate {
    std::mutex mu;
    std::condition_variable cv;
    bool open = false;
    void release() {
        {
            std::lock_guard<std::mutex> lk(mu);
            open = true;
        }
        cv.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return open; });
    }
};

struct sipp_prof_entry {
    int calls = 0;
    double ms = 0.0;
};

struct sipp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    sipp_stark_config cfg{};
    char err[512] = {0};

    // bump arena (device)
    char* arena = nullptr;
    size_t arena_size = 0;
    size_t arena_off = 0;
    size_t arena_peak = 0;

    // persistent small tables (device), keyed by (kind, a, b)
    std::map<std::tuple<int, uint64_t, uint64_t>, uint64_t*> tables;

    // profiling
    bool prof = false;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    std::vector<std::tuple<std::string, size_t, size_t>> pending;  // name, ev index a, b
    std::map<std::string, sipp_prof_entry> prof_acc;
    hipEvent_t t0 = nullptr, t1 = nullptr;

    // pinned host staging
    uint64_t* h_pinned = nullptr;
    size_t h_pinned_words = 0;

    // set by sipp_instance_prove for the duration of one proof (see sipp_gate)
    sipp_gate* gate_wait = nullptr;     // wait for it before the first launch
    sipp_gate* gate_release = nullptr;  // released after the trace fill (and on every exit path)

    // second stream for the G1 half of sipp_fold_outputs (created on first use)
    hipStream_t aux_stream = nullptr;
    // sipp_fold_begin / sipp_fold_finish: a third stream (G2 chain) and the state that lives between the two calls
    hipStream_t aux2_stream = nullptr;
    struct Fold {
        bool active = false;
        size_t mark = 0;
        uint32_t* d_ios[2] = {nullptr, nullptr};
        void* rows[2] = {nullptr, nullptr};
        int* d_err = nullptr;
        uint32_t num_io[2] = {0, 0};   // padded
        size_t n_in[2] = {0, 0};
    } fold;

    // sipp_exp_outputs: sipp_trace_fill stops after the accumulator chains and writes the outputs into the records
    bool outputs_only = false;
    // sipp_ctx_set_hardened: kinds 0 / 1 on this ctx mean the hardened G1 / G2 AIRs (kinds 4 / 5)
    bool hardened = false;
    uint32_t kernel_routes = 0;     // sipp_ctx_set_kernel_routes: SIPP_ROUTE_* bits (fallback kernels kept under test)
    // witness.hip: the launch sequence of sipp_plonk_generate_witness_levels captured as a hipGraph (two launches per level: launch-bound),
    // replayed while the key (buffers, schedule, generators) stays the same; released by sipp_witness_graph_release
    struct WitnessGraph {
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        std::vector<uint64_t> key;
    } wgraph;

    // sipp_prove_async / sipp_wait: one worker thread per ctx, started on first use, one job at a time
    struct Async {
        std::thread th;
        std::mutex mu;
        std::condition_variable cv;
        bool started = false, has_job = false, done = false, quit = false;
        int kind = 0;
        const uint32_t* ios = nullptr;
        size_t num_io = 0, cap = 0, len = 0;
        uint64_t* out = nullptr;
        int rc = 0;
    } async;
};

#define SIPP_CHECK_HIP(ctx, expr)                                                                   \
    do {                                                                                            \
        hipError_t e__ = (expr);                                                                    \
        if (e__ != hipSuccess) {                                                                    \
            snprintf((ctx)->err, sizeof((ctx)->err), "%s:%d: %s -> %s", __FILE__, __LINE__, #expr,  \
                     hipGetErrorString(e__));                                                       \
            return SIPP_E_HIP;                                                                      \
        }                                                                                           \
    } while (0)

#define SIPP_TRY(expr)              \
    do {                            \
        int rc__ = (expr);          \
        if (rc__ != SIPP_OK) return rc__; \
    } while (0)

static inline int sipp_fail(sipp_ctx* ctx, int code, const char* msg) {
    if (ctx) snprintf(ctx->err, sizeof(ctx->err), "%s", msg);
    return code;
}

// ---- arena -------------------------------------------------------------------
struct ArenaMark {
    size_t off;
};
static inline ArenaMark arena_mark(sipp_ctx* ctx) { return ArenaMark{ctx->arena_off}; }
static inline void arena_release(sipp_ctx* ctx, ArenaMark m) { ctx->arena_off = m.off; }
// returns nullptr when exhausted
static inline void* arena_alloc(sipp_ctx* ctx, size_t bytes) {
    size_t off = (ctx->arena_off + 255) & ~(size_t)255;
    if (off + bytes > ctx->arena_size) {
        snprintf(ctx->err, sizeof(ctx->err), "workspace arena exhausted: need %zu more bytes (size %zu, used %zu)",
                 bytes, ctx->arena_size, off);
        return nullptr;
    }
    ctx->arena_off = off + bytes;
    if (ctx->arena_off > ctx->arena_peak) ctx->arena_peak = ctx->arena_off;
    return ctx->arena + off;
}
// releases everything allocated after its construction on EVERY exit path (error returns included)
struct ArenaScope {
    sipp_ctx* ctx;
    ArenaMark mark;
    explicit ArenaScope(sipp_ctx* c) : ctx(c), mark(arena_mark(c)) {}
    ~ArenaScope() { arena_release(ctx, mark); }
    ArenaScope(const ArenaScope&) = delete;
    ArenaScope& operator=(const ArenaScope&) = delete;
};
template <typename T>
static inline T* arena_alloc_t(sipp_ctx* ctx, size_t count) {
    return reinterpret_cast<T*>(arena_alloc(ctx, count * sizeof(T)));
}

// ---- profiling-aware launch bracket ------------------------------------------
// (Round 2 had a SIPP_SYNC_SPIN knob that replaced hipStreamSynchronize by a hipStreamQuery poll through a macro of the same name:
// it measured inside the box-to-box spread -- 61.5-61.7 against 61.8-62.0 ms -- burned a core per ctx and shadowed a HIP API name;
// removed.)
struct ProfScope {
    sipp_ctx* ctx;
    const char* name;
    size_t ia = 0, ib = 0;
    bool on;
    ProfScope(sipp_ctx* c, const char* n) : ctx(c), name(n), on(c->prof) {
        if (!on) return;
        if (ctx->ev_used + 2 > ctx->ev_pool.size()) {
            size_t old = ctx->ev_pool.size();
            ctx->ev_pool.resize(old + 256);
            for (size_t i = old; i < ctx->ev_pool.size(); i++) (void)hipEventCreate(&ctx->ev_pool[i]);
        }
        ia = ctx->ev_used++;
        ib = ctx->ev_used++;
        (void)hipEventRecord(ctx->ev_pool[ia], ctx->stream);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(ctx->ev_pool[ib], ctx->stream);
        ctx->pending.emplace_back(name, ia, ib);
    }
};

// ---- persistent tables ---------------------------------------------------------
enum TableKind {
    TAB_WR = 1,       // w_R^x, x < R/2            key (logR, inverse)
    TAB_TW_LO = 2,    // w_m^x, x < 2^L            key (logm, inverse)
    TAB_TW_HI = 3,    // w_m^(x * 2^L)             key (logm, inverse)
    TAB_POW_LO = 4,   // base^x * c                key (id, L)
    TAB_POW_HI = 5,   // base^(x * 2^L)            key (id, L)
};

uint64_t* sipp_table_get(sipp_ctx* ctx, int kind, uint64_t a, uint64_t b);
int sipp_table_put(sipp_ctx* ctx, int kind, uint64_t a, uint64_t b, const std::vector<uint64_t>& host, uint64_t** out);

// ---- internal kernels' host entry points (ntt.hip / poseidon.hip) ---------------
// Direction of an in-place radix-2 transform over a column-major batch.
//   DIF: natural order in  -> bit-reversed order out
//   DIT: bit-reversed in   -> natural order out
struct NttDiag {
    // optional per-element scaling by base^(natural index) * c, applied on the natural-order side
    // (DIF: input, DIT: output).  base == 0 disables; c == 0 means 1.
    uint64_t base = 0;
    uint64_t c = 0;
};
int sipp_ntt_dif(sipp_ctx* ctx, const uint64_t* d_in, size_t in_stride, uint32_t log_n_in, uint64_t* d_out,
                 size_t out_stride, uint32_t log_n, size_t ncols, bool inverse, NttDiag diag);
int sipp_ntt_dit(sipp_ctx* ctx, uint64_t* d_io, size_t stride, uint32_t log_n, size_t ncols, bool inverse,
                 NttDiag diag);
// fused PolynomialBatch transforms (ntt.hip); SIPP_E_UNSUPPORTED = shape not covered, use the pass-by-pass path
int sipp_lde_from_values(sipp_ctx* ctx, const uint64_t* d_values, uint64_t* d_coeffs, uint64_t* d_lde, size_t ncols, uint32_t log_n,
                         uint32_t rate_bits);
int sipp_lde_from_coeffs(sipp_ctx* ctx, const uint64_t* d_coeffs, uint64_t* d_lde, size_t ncols, uint32_t log_n, uint32_t rate_bits);
// the same two for long columns (ntt_tree.hip: tree-of-rings transforms, no diagonals, no bit-reversal copy)
bool sipp_tree_ntt_enabled(uint32_t log_n);
int sipp_tree_lde_from_values(sipp_ctx* ctx, const uint64_t* d_values, uint64_t* d_coeffs, uint64_t* d_lde, size_t ncols,
                              uint32_t log_n, uint32_t rate_bits);
int sipp_tree_lde_from_coeffs(sipp_ctx* ctx, const uint64_t* d_coeffs, uint64_t* d_lde, size_t ncols, uint32_t log_n,
                              uint32_t rate_bits);
int sipp_tree_coset_eval(sipp_ctx* ctx, const uint64_t* d_coeffs, uint64_t* d_out, size_t ncols, uint32_t log_n, uint32_t log_m);
// out[c][j] = in[c][bitrev(j)]  (out != in)
int sipp_bitrev_cols(sipp_ctx* ctx, const uint64_t* d_in, size_t in_stride, uint64_t* d_out, size_t out_stride,
                     uint32_t log_n, size_t ncols);

int sipp_k_poseidon_leaves(sipp_ctx* ctx, const uint64_t* d_lde, size_t col_stride, size_t ncols, uint32_t log_leaves,
                           uint64_t* d_digests);
int sipp_k_merkle_levels(sipp_ctx* ctx, uint64_t* d_tree, uint32_t log_leaves, uint32_t cap_height);
int sipp_k_poseidon_permute(sipp_ctx* ctx, uint64_t* d_states, size_t n);
void sipp_witness_graph_release(sipp_ctx* ctx);   // witness.hip
int sipp_plonk_circuit_check(sipp_ctx* ctx, const sipp_plonk_circuit* c, const sipp_plonk_params* p);   // plonk.hip

// ---- AIR layer (trace.hip / quotient.hip / stark.hip) -----------------------------------------
#include "air_tables.h"
const air_spec_t* sipp_air_get(int kind, uint32_t log_n);
const int64_t* sipp_air_prog_device(sipp_ctx* ctx, const air_spec_t* a);
int sipp_trace_fill(sipp_ctx* ctx, const air_spec_t* a, const uint32_t* d_ios, uint32_t num_io, uint32_t log_n,
                    uint64_t* d_trace, int* d_err);
size_t sipp_curve_rows_bytes(int kind, uint32_t log_n);
// mapg2.hip: the map Fp2 -> E'(Fp2) (MapToG2 AIR, kind 3): primary witness rows / (x, y) written into the records
int sipp_mapg2_fill(sipp_ctx* ctx, const air_spec_t* a, const uint32_t* d_ios, uint32_t num_io, uint32_t log_n, uint64_t* d_trace,
                    int* d_err);
int sipp_mapg2_outputs(sipp_ctx* ctx, uint32_t* d_ios, uint32_t num_io, int* d_err);
// pairing.hip: the final pairing (pairing AIR, kind 6): primary witness rows, or (ctx->outputs_only) Z written into the records
int sipp_pairing_fill(sipp_ctx* ctx, const air_spec_t* a, const uint32_t* d_ios, uint32_t num_io, uint32_t log_n, uint64_t* d_trace,
                      int* d_err);
// the AIR's selector columns (value-periodic, small integers) on the device: [n_vflag][2^log_rows] int8; nullptr for an AIR without
const int8_t* sipp_air_vflag_device(sipp_ctx* ctx, const air_spec_t* a);
// outputs of n1 G1 and n2 G2 obligations, the two accumulator chains on two streams (native.hip's fold of a SIPP round)
int sipp_fold_outputs(sipp_ctx* ctx, uint32_t* g1_ios, size_t n1, uint32_t* g2_ios, size_t n2);
// the same in two phases: begin starts the 255 doublings of every record's point x on two side streams (the exponent and
// output words of the records are ignored), finish uploads the complete records, selects and sums the powers, and writes the
// outputs into the records.  Between the two calls the ctx's main stream is free (native.hip computes the round's pairing
// products there); nothing else may allocate BELOW the arena mark taken by begin.
int sipp_fold_begin(sipp_ctx* ctx, const uint32_t* g1_ios, size_t n1, const uint32_t* g2_ios, size_t n2);
int sipp_fold_finish(sipp_ctx* ctx, uint32_t* g1_ios, size_t n1, uint32_t* g2_ios, size_t n2);
size_t sipp_fold_rows_bytes(int kind, uint32_t num_io);
int sipp_fold_chain_begin(sipp_ctx* ctx, int kind, const uint32_t* d_ios, uint32_t num_io, uint32_t ppi, void* rows, hipStream_t st);
int sipp_fold_chain_finish(sipp_ctx* ctx, int kind, uint32_t* d_ios, uint32_t num_io, uint32_t ppi, void* rows, int* d_err,
                           hipStream_t st);
