// sipp_amd/csrc/api.hip -- C-ABI entry points of include/sipp_hip.h: context, arena, tables,
// profiling and the building-block calls (NTT / LDE / Poseidon leaves / Merkle / commit).
#include <algorithm>
#include <string>

#include "ctx.hpp"
#include "host_poseidon.hpp"

int sipp_poseidon_init_constants(sipp_ctx* ctx);  // poseidon.hip

extern "C" {

void sipp_default_config(sipp_stark_config* cfg) {
    if (!cfg) return;
    cfg->rate_bits = 1;
    cfg->cap_height = 4;
    cfg->pow_bits = 16;
    cfg->arity_bits = 4;
    cfg->final_poly_bits = 5;
    cfg->num_queries = 84;
    cfg->num_challenges = 2;
    cfg->pow_rule = SIPP_POW_DUPLEX;
    cfg->fs_rule = SIPP_FS_STATEMENT;
    cfg->lookup_rule = SIPP_LOOKUP_INDEPENDENT;
}

// The stream of a ctx.  level > 0: a high-priority stream.  Otherwise a stream with a HARDWARE QUEUE OF ITS OWN: the runtime
// multiplexes ordinary streams onto a small pool of queues per priority (GPU_MAX_HW_QUEUES, 4), a stream created with a CU mask gets
// a queue to itself -- the mask here is all ones, every CU stays usable.  Measured on the n = 128 instance (HISTORY.md section 6c):
// G1 (formerly on the low-priority pool) on its own queue is worth 1.5 ms of 60 single and 0.9 of 51.8 ms queued; the blocking flag,
// the priority value and the number of pool queues are not what moves it.  Such a stream has normal priority (the call takes none),
// so "low" and "normal" are the same thing now.  SIPP_DEDICATED_QUEUES=0 restores pool streams with priorities.
static hipError_t create_ctx_stream(int device, int level, hipStream_t* out) {
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);  // lo = least, hi = greatest priority (numerically lower)
    static const int dedicated = [] { const char* e = getenv("SIPP_DEDICATED_QUEUES"); return e ? atoi(e) : 1; }();
    if (level <= 0 && dedicated) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0 && prop.multiProcessorCount <= 1024) {
            uint32_t mask[32];
            const uint32_t words = ((uint32_t)prop.multiProcessorCount + 31) / 32;
            for (uint32_t i = 0; i < words; i++) {
                const uint32_t left = (uint32_t)prop.multiProcessorCount - 32 * i;
                mask[i] = left >= 32 ? 0xffffffffu : (1u << left) - 1;
            }
            if (hipExtStreamCreateWithCUMask(out, words, mask) == hipSuccess) return hipSuccess;
            (void)hipGetLastError();   // no queue left for it: a pool stream below
        }
    }
    return hipStreamCreateWithPriority(out, hipStreamNonBlocking, level > 0 ? hi : level < 0 ? lo : 0);
}

int sipp_ctx_create(sipp_ctx** out, int device, const sipp_stark_config* cfg, size_t workspace_bytes) {
    if (!out) return SIPP_E_BADARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return SIPP_E_HIP;
    sipp_ctx* ctx = new sipp_ctx();
    ctx->device = device;
    if (cfg)
        ctx->cfg = *cfg;
    else
        sipp_default_config(&ctx->cfg);
    // blowup 2 / 4 / 8, reduction arity 2 / 4 / 8 / 16 (constant), two challenges (the AIRs' quotient kernels fold exactly two)
    if (ctx->cfg.rate_bits < 1 || ctx->cfg.rate_bits > 3 || ctx->cfg.num_challenges != 2 || ctx->cfg.arity_bits < 1 ||
        ctx->cfg.arity_bits > 4 || ctx->cfg.cap_height > 8 || ctx->cfg.pow_bits > 32 || ctx->cfg.pow_rule > SIPP_POW_HASH ||
        ctx->cfg.fs_rule > SIPP_FS_UPSTREAM || ctx->cfg.lookup_rule > SIPP_LOOKUP_SHARED || ctx->cfg.num_queries == 0 || ctx->cfg.num_queries > 1024 || ctx->cfg.final_poly_bits > 12) {
        delete ctx;
        return SIPP_E_UNSUPPORTED;
    }
    auto bail = [&](int rc) {
        // no ctx survives to carry the message: say it on stderr (creation failures are configuration errors)
        hipError_t last = hipGetLastError();
        fprintf(stderr, "sipp_ctx_create(device %d, workspace %zu bytes) failed with status %d (last HIP error: %s)\n", device,
                workspace_bytes, rc, hipGetErrorString(last));
        sipp_ctx_destroy(ctx);
        return rc;
    };
    if (hipSetDevice(device) != hipSuccess) return bail(SIPP_E_HIP);
    // level 0; sipp_ctx_set_stream_priority replaces the stream (one ctx per STARK: the thin, latency-bound proof goes ahead)
    if (create_ctx_stream(device, 0, &ctx->stream) != hipSuccess) return bail(SIPP_E_HIP);
    if (workspace_bytes == 0) workspace_bytes = (size_t)24 << 30;
    ctx->arena_size = workspace_bytes;
    if (hipMalloc((void**)&ctx->arena, workspace_bytes) != hipSuccess) return bail(SIPP_E_NOMEM);
    if (hipEventCreate(&ctx->t0) != hipSuccess || hipEventCreate(&ctx->t1) != hipSuccess) return bail(SIPP_E_HIP);
    ctx->h_pinned_words = (size_t)1 << 23;  // 64 MiB of pinned staging for IO records / query rows
    if (hipHostMalloc((void**)&ctx->h_pinned, ctx->h_pinned_words * 8) != hipSuccess) return bail(SIPP_E_NOMEM);
    int rc = sipp_poseidon_init_constants(ctx);
    if (rc != SIPP_OK) return bail(rc);
    *out = ctx;
    return SIPP_OK;
}

void sipp_ctx_destroy(sipp_ctx* ctx) {
    if (!ctx) return;
    if (ctx->async.started) {
        // a proof still in flight runs to its end (its buffers belong to the caller until then), then the worker exits
        {
            std::unique_lock<std::mutex> lk(ctx->async.mu);
            ctx->async.cv.wait(lk, [&] { return !ctx->async.has_job || ctx->async.done; });
            ctx->async.quit = true;
        }
        ctx->async.cv.notify_all();
        ctx->async.th.join();
    }
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    sipp_witness_graph_release(ctx);
    for (auto& kv : ctx->tables) (void)hipFree(kv.second);
    for (auto e : ctx->ev_pool) (void)hipEventDestroy(e);
    if (ctx->t0) (void)hipEventDestroy(ctx->t0);
    if (ctx->t1) (void)hipEventDestroy(ctx->t1);
    if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
    if (ctx->arena) (void)hipFree(ctx->arena);
    if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);
    if (ctx->aux2_stream) (void)hipStreamDestroy(ctx->aux2_stream);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

uint32_t sipp_abi_version(void) { return SIPP_ABI_VERSION; }

int sipp_ctx_create_checked(sipp_ctx** out, int device, const sipp_stark_config* cfg, size_t cfg_size, uint32_t abi_version,
                            size_t workspace_bytes) {
    if (out) *out = nullptr;
    // a caller built against another header (a shorter sipp_stark_config would be read past its end) is refused before anything is read
    if (abi_version != SIPP_ABI_VERSION || (cfg && cfg_size != sizeof(sipp_stark_config))) return SIPP_E_BADARG;
    return sipp_ctx_create(out, device, cfg, workspace_bytes);
}

int sipp_ctx_set_kernel_routes(sipp_ctx* ctx, uint32_t routes) {
    if (!ctx || (routes & ~(uint32_t)(SIPP_ROUTE_OPENINGS_UNGROUPED | SIPP_ROUTE_LDE_COLUMN_WIDE | SIPP_ROUTE_WITNESS_NO_GRAPH))) return SIPP_E_BADARG;
    std::unique_lock<std::mutex> lk(ctx->async.mu);
    if (ctx->async.has_job) {
        lk.unlock();
        return sipp_fail(ctx, SIPP_E_BADARG, "set_kernel_routes: a proof is in flight on this ctx");
    }
    ctx->kernel_routes = routes;
    return SIPP_OK;
}

int sipp_ctx_set_hardened(sipp_ctx* ctx, int on) {
    if (!ctx) return SIPP_E_BADARG;
    std::unique_lock<std::mutex> lk(ctx->async.mu);
    if (ctx->async.has_job) {
        lk.unlock();
        return sipp_fail(ctx, SIPP_E_BADARG, "set_hardened: a proof is in flight on this ctx");
    }
    ctx->hardened = on != 0;
    return SIPP_OK;
}

int sipp_ctx_set_stream_priority(sipp_ctx* ctx, int level) {
    if (!ctx) return SIPP_E_BADARG;
    {
        std::unique_lock<std::mutex> lk(ctx->async.mu);
        if (ctx->async.has_job) {
            lk.unlock();
            return sipp_fail(ctx, SIPP_E_BADARG, "set_stream_priority: a proof is in flight on this ctx");
        }
    }
    SIPP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t ns = nullptr;
    if (create_ctx_stream(ctx->device, level, &ns) != hipSuccess) return sipp_fail(ctx, SIPP_E_HIP, "set_stream_priority: no stream");
    if (ctx->stream) {
        (void)hipStreamSynchronize(ctx->stream);
        if (ctx->prof) {  // pending event pairs belong to the old stream: fold them into the totals first
            (void)sipp_profile_enable(ctx, 0);
            ctx->prof = true;
        }
        (void)hipStreamDestroy(ctx->stream);
    }
    ctx->stream = ns;
    return SIPP_OK;
}

const char* sipp_last_error(const sipp_ctx* ctx) { return ctx ? ctx->err : "null ctx"; }

int sipp_sync(sipp_ctx* ctx) {
    if (!ctx) return SIPP_E_BADARG;
    SIPP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    SIPP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SIPP_OK;
}

void* sipp_stream(sipp_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

// ---- profiling ------------------------------------------------------------------
static void prof_drain(sipp_ctx* ctx) {
    if (ctx->pending.empty()) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& p : ctx->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ctx->ev_pool[std::get<1>(p)], ctx->ev_pool[std::get<2>(p)]) == hipSuccess) {
            auto& e = ctx->prof_acc[std::get<0>(p)];
            e.calls++;
            e.ms += ms;
        }
    }
    ctx->pending.clear();
    ctx->ev_used = 0;
}

int sipp_profile_enable(sipp_ctx* ctx, int enable) {
    if (!ctx) return SIPP_E_BADARG;
    if (!enable) prof_drain(ctx);
    ctx->prof = enable != 0;
    return SIPP_OK;
}

int sipp_profile_reset(sipp_ctx* ctx) {
    if (!ctx) return SIPP_E_BADARG;
    prof_drain(ctx);
    ctx->prof_acc.clear();
    return SIPP_OK;
}

int sipp_profile_report(sipp_ctx* ctx, char* buf, size_t cap) {
    if (!ctx || !buf || cap < 3) return SIPP_E_BADARG;
    prof_drain(ctx);
    std::string s = "{";
    bool first = true;
    for (auto& kv : ctx->prof_acc) {
        char line[256];
        snprintf(line, sizeof line, "%s\"%s\": {\"calls\": %d, \"ms\": %.6f}", first ? "" : ", ", kv.first.c_str(),
                 kv.second.calls, kv.second.ms);
        s += line;
        first = false;
    }
    s += "}";
    if (s.size() + 1 > cap) return SIPP_E_BUFSZ;
    memcpy(buf, s.c_str(), s.size() + 1);
    return SIPP_OK;
}

int sipp_timer_start(sipp_ctx* ctx) {
    if (!ctx) return SIPP_E_BADARG;
    SIPP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    SIPP_CHECK_HIP(ctx, hipEventRecord(ctx->t0, ctx->stream));
    return SIPP_OK;
}

int sipp_timer_stop(sipp_ctx* ctx, float* ms) {
    if (!ctx || !ms) return SIPP_E_BADARG;
    SIPP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    SIPP_CHECK_HIP(ctx, hipEventRecord(ctx->t1, ctx->stream));
    SIPP_CHECK_HIP(ctx, hipEventSynchronize(ctx->t1));
    SIPP_CHECK_HIP(ctx, hipEventElapsedTime(ms, ctx->t0, ctx->t1));
    return SIPP_OK;
}

// ---- building blocks --------------------------------------------------------------
int sipp_ntt_batch(sipp_ctx* ctx, uint64_t* d_cols, size_t col_stride, size_t ncols, uint32_t log_n, int inverse) {
    if (!ctx || !d_cols) return SIPP_E_BADARG;
    SIPP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    size_t n = (size_t)1 << log_n;
    if (col_stride < n) return sipp_fail(ctx, SIPP_E_BADARG, "ntt_batch: col_stride < n");
    ArenaScope scope(ctx);
    uint64_t* tmp = arena_alloc_t<uint64_t>(ctx, n * ncols);
    if (!tmp) return SIPP_E_NOMEM;
    // natural -> (DIF) bit-reversed -> permuted back to natural
    int rc = sipp_ntt_dif(ctx, d_cols, col_stride, log_n, tmp, n, log_n, ncols, inverse != 0, NttDiag{});
    if (rc == SIPP_OK) rc = sipp_bitrev_cols(ctx, tmp, n, d_cols, col_stride, log_n, ncols);
    const int rs = sipp_sync(ctx);   // the scratch is handed back only after the kernels are done with it
    return rc == SIPP_OK ? rs : rc;
}

// values (natural) -> coeffs (natural) + LDE (leaf order)
static int lde_from_values(sipp_ctx* ctx, const uint64_t* d_values, uint64_t* d_coeffs, uint64_t* d_lde, size_t ncols,
                           uint32_t log_n) {
    const size_t n = (size_t)1 << log_n;
    const uint32_t rb = ctx->cfg.rate_bits;
    SIPP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    ArenaScope scope(ctx);   // released on every exit path; the stream is ordered, so later users of the block wait
    uint64_t* src = const_cast<uint64_t*>(d_values);
    int rc = sipp_lde_from_values(ctx, d_values, d_coeffs, d_lde, ncols, log_n, rb);
    if (rc != SIPP_E_UNSUPPORTED) return rc;
    rc = SIPP_OK;
    if (d_values == d_coeffs) {
        uint64_t* tmp = arena_alloc_t<uint64_t>(ctx, n * ncols);
        if (!tmp) return SIPP_E_NOMEM;
        SIPP_CHECK_HIP(ctx, hipMemcpyAsync(tmp, d_values, n * ncols * 8, hipMemcpyDeviceToDevice, ctx->stream));
        src = tmp;
    }
    rc = sipp_bitrev_cols(ctx, src, n, d_coeffs, n, log_n, ncols);
    if (rc == SIPP_OK) rc = sipp_ntt_dit(ctx, d_coeffs, n, log_n, ncols, /*inverse=*/true, NttDiag{});
    if (rc == SIPP_OK)
        rc = sipp_ntt_dif(ctx, d_coeffs, n, log_n, d_lde, n << rb, log_n + rb, ncols, false, NttDiag{gl::GEN, 0});
    return rc;
}

int sipp_lde_batch(sipp_ctx* ctx, const uint64_t* d_values, uint64_t* d_coeffs, uint64_t* d_lde, size_t ncols,
                   uint32_t log_n) {
    if (!ctx || !d_values || !d_coeffs || !d_lde) return SIPP_E_BADARG;
    SIPP_TRY(lde_from_values(ctx, d_values, d_coeffs, d_lde, ncols, log_n));
    return sipp_sync(ctx);
}

int sipp_poseidon_leaves(sipp_ctx* ctx, const uint64_t* d_lde, size_t ncols, uint32_t log_leaves,
                         uint64_t* d_digests) {
    if (!ctx || !d_lde || !d_digests) return SIPP_E_BADARG;
    SIPP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    SIPP_TRY(sipp_k_poseidon_leaves(ctx, d_lde, (size_t)1 << log_leaves, ncols, log_leaves, d_digests));
    return sipp_sync(ctx);
}

static int read_cap(sipp_ctx* ctx, const uint64_t* d_tree, uint32_t log_leaves, uint64_t* cap_out) {
    uint32_t ch = std::min(ctx->cfg.cap_height, log_leaves);
    uint64_t off = 0;
    for (uint32_t l = 0; l < log_leaves - ch; l++) off += (uint64_t)1 << (log_leaves - l);
    SIPP_CHECK_HIP(ctx, hipMemcpyAsync(cap_out, d_tree + 4 * off, ((size_t)4 << ch) * 8, hipMemcpyDeviceToHost,
                                       ctx->stream));
    return sipp_sync(ctx);
}

int sipp_merkle_cap(sipp_ctx* ctx, uint64_t* d_tree, uint32_t log_leaves, uint64_t* cap_out) {
    if (!ctx || !d_tree || !cap_out) return SIPP_E_BADARG;
    SIPP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    SIPP_TRY(sipp_k_merkle_levels(ctx, d_tree, log_leaves, ctx->cfg.cap_height));
    return read_cap(ctx, d_tree, log_leaves, cap_out);
}

int sipp_commit_batch(sipp_ctx* ctx, const uint64_t* d_values, uint64_t* d_coeffs, uint64_t* d_lde, uint64_t* d_tree,
                      size_t ncols, uint32_t log_n, uint64_t* cap_out) {
    if (!ctx || !d_values || !d_coeffs || !d_lde || !d_tree || !cap_out) return SIPP_E_BADARG;
    const uint32_t log_m = log_n + ctx->cfg.rate_bits;
    SIPP_TRY(lde_from_values(ctx, d_values, d_coeffs, d_lde, ncols, log_n));
    SIPP_TRY(sipp_k_poseidon_leaves(ctx, d_lde, (size_t)1 << log_m, ncols, log_m, d_tree));
    SIPP_TRY(sipp_k_merkle_levels(ctx, d_tree, log_m, ctx->cfg.cap_height));
    return read_cap(ctx, d_tree, log_m, cap_out);
}

int sipp_host_poseidon_permute(uint64_t* states, size_t n, int impl) {
    if (!states && n) return SIPP_E_BADARG;
    for (size_t i = 0; i < n; i++) {
        if (impl < 0) {
            host::poseidon_permute(states + 12 * i);
            continue;
        }
        const int r = host::poseidon_permute_impl(states + 12 * i, impl);
        if (r == -1) return SIPP_E_UNSUPPORTED;
        if (r != 0 || impl >= 10) return SIPP_E_BADARG;
    }
    return SIPP_OK;
}

int sipp_poseidon_permute(sipp_ctx* ctx, uint64_t* d_states, size_t n) {
    if (!ctx || !d_states) return SIPP_E_BADARG;
    SIPP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    SIPP_TRY(sipp_k_poseidon_permute(ctx, d_states, n));
    return sipp_sync(ctx);
}

// ---- provers: implemented in stark.hip -----------------------------------------------
}  // extern "C"

// ---- persistent tables ------------------------------------------------------------------
uint64_t* sipp_table_get(sipp_ctx* ctx, int kind, uint64_t a, uint64_t b) {
    auto it = ctx->tables.find(std::make_tuple(kind, a, b));
    return it == ctx->tables.end() ? nullptr : it->second;
}

int sipp_table_put(sipp_ctx* ctx, int kind, uint64_t a, uint64_t b, const std::vector<uint64_t>& host,
                   uint64_t** out) {
    uint64_t* d = nullptr;
    SIPP_CHECK_HIP(ctx, hipSetDevice(ctx->device));   // the table must live on the ctx's GPU whatever the thread last used
    // tables are created once per (kind, size) and live as long as the ctx: the only hipMalloc outside
    // sipp_ctx_create, and only on the first use of a new transform size.
    SIPP_CHECK_HIP(ctx, hipMalloc((void**)&d, host.size() * 8));
    hipError_t e = hipMemcpy(d, host.data(), host.size() * 8, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(d);
        SIPP_CHECK_HIP(ctx, e);
    }
    ctx->tables[std::make_tuple(kind, a, b)] = d;
    *out = d;
    return SIPP_OK;
}
