// sipp_amd/csrc/stark.hip -- the three provers behind reference src/verifier_circuit.rs:133-135.
#include "ctx.hpp"

extern "C" {

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
int sipp_stark_shape(const sipp_ctx*, int, size_t, uint32_t*, uint32_t*, uint32_t*, uint32_t*) {
    return SIPP_E_UNSUPPORTED;
}
}
