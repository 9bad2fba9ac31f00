// sipp_amd/csrc/stark.hip -- the three provers behind reference src/verifier_circuit.rs:133-135.
#include "ctx.hpp"

static const int IO_WORDS[3] = {SIPP_G1_IO_WORDS, SIPP_G2_IO_WORDS, SIPP_FQ12_IO_WORDS};

struct Shape {
    const sipp_air_t* air;
    uint32_t log_n;
    uint32_t num_io;  // padded
    int W, P, Q;
};

static int shape_of(int kind, size_t num_io, Shape* s) {
    if (kind < 0 || kind > 2 || num_io == 0 || num_io > ((size_t)1 << 17)) return SIPP_E_BADARG;
    uint32_t nio = 1;
    while (nio < num_io) nio <<= 1;
    uint32_t log_n = 9;
    while ((1u << (log_n - 9)) < nio) log_n++;
    s->air = sipp_air_get(kind, log_n);
    if (!s->air) return SIPP_E_UNSUPPORTED;
    s->log_n = log_n;
    s->num_io = nio;
    s->W = s->air->n_main + 2 * s->air->n_checked;
    s->P = 2 * s->air->n_checked;
    s->Q = 4;
    return SIPP_OK;
}

// pads the IO list with copies of the last record and uploads it
static int upload_ios(sipp_ctx* ctx, int kind, const uint32_t* ios, size_t num_io, const Shape& s, uint32_t** d_ios) {
    const size_t ppi = IO_WORDS[kind];
    const size_t words = (size_t)s.num_io * ppi;
    if (words * 4 > ctx->h_pinned_words * 8) return sipp_fail(ctx, SIPP_E_NOMEM, "IO list larger than the pinned staging buffer");
    uint32_t* h = reinterpret_cast<uint32_t*>(ctx->h_pinned);
    for (size_t io = 0; io < s.num_io; io++)
        memcpy(h + io * ppi, ios + (io < num_io ? io : num_io - 1) * ppi, ppi * 4);
    *d_ios = arena_alloc_t<uint32_t>(ctx, words);
    if (!*d_ios) return SIPP_E_NOMEM;
    SIPP_CHECK_HIP(ctx, hipMemcpyAsync(*d_ios, h, words * 4, hipMemcpyHostToDevice, ctx->stream));
    return SIPP_OK;
}

extern "C" {

int sipp_stark_shape(const sipp_ctx* ctx, int kind, size_t num_io, uint32_t* log_rows, uint32_t* main_cols,
                     uint32_t* perm_cols, uint32_t* quotient_cols) {
    (void)ctx;
    Shape s;
    SIPP_TRY(shape_of(kind, num_io, &s));
    if (log_rows) *log_rows = s.log_n;
    if (main_cols) *main_cols = (uint32_t)s.W;
    if (perm_cols) *perm_cols = (uint32_t)s.P;
    if (quotient_cols) *quotient_cols = (uint32_t)s.Q;
    return SIPP_OK;
}

int sipp_trace_build(sipp_ctx* ctx, int kind, const uint32_t* ios, size_t num_io, uint64_t* d_trace) {
    if (!ctx || !ios || !d_trace) return SIPP_E_BADARG;
    Shape s;
    SIPP_TRY(shape_of(kind, num_io, &s));
    ArenaMark m = arena_mark(ctx);
    uint32_t* d_ios = nullptr;
    int rc = upload_ios(ctx, kind, ios, num_io, s, &d_ios);
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

int sipp_g1_exp_prove(sipp_ctx* ctx, const uint32_t*, size_t, uint64_t*, size_t, size_t*) {
    return sipp_fail(ctx, SIPP_E_UNSUPPORTED, "g1_exp_prove: not built yet");
}
int sipp_g2_exp_prove(sipp_ctx* ctx, const uint32_t*, size_t, uint64_t*, size_t, size_t*) {
    return sipp_fail(ctx, SIPP_E_UNSUPPORTED, "g2_exp_prove: not built yet");
}
int sipp_fq12_exp_prove(sipp_ctx* ctx, const uint32_t*, size_t, uint64_t*, size_t, size_t*) {
    return sipp_fail(ctx, SIPP_E_UNSUPPORTED, "fq12_exp_prove: not built yet");
}
size_t sipp_proof_size(const sipp_ctx*, int, size_t) { return 0; }
}
