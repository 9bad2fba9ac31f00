// sipp_amd/csrc/circuit_data.hip -- plonky2's CircuitData for the outer proof behind HOST pointers: what the reference does at
// src/verifier_circuit.rs:225 (`builder.build::<C>()`: constants_sigmas committed once, the generators ordered once), :253 (`data.prove(pw)`)
// and :254 (`data.verify(proof)`), for a caller that holds no device memory of its own (the Rust shim; include/sipp_host.hpp's CircuitData).
// Nothing new is computed here: build = sipp_commit_batch_ex over the uploaded constants_sigmas; prove = upload of the wire table with its
// input cells, sipp_plonk_generate_witness[_levels], sipp_plonk_prove_gates; verify = sipp_plonk_verify_gates with the data's own cap / digest.
// Device memory of a circuit data is its own (hipMalloc at build: the arena of the ctx is the provers' scratch and empties after each proof).
#include "ctx.hpp"
#include "host_challenger.hpp"

struct sipp_circuit_data {
    sipp_ctx* ctx = nullptr;
    uint32_t log_n = 0;
    sipp_plonk_params p{};
    sipp_fri_params fp{};
    sipp_plonk_circuit circ{};
    std::vector<sipp_plonk_gate> gates;
    std::vector<int64_t> programs;
    std::vector<sipp_plonk_generator> gens;
    bool has_sched = false;
    sipp_plonk_schedule sched{};
    std::vector<uint32_t> level_offsets, copy_offsets;
    std::vector<void*> dev;                 // everything hipMalloc'ed for this circuit
    uint64_t *d_cs = nullptr, *d_wires = nullptr;
    sipp_oracle cs_oracle{};
    std::vector<uint64_t> cap;
    uint64_t digest[4] = {0, 0, 0, 0};
    ~sipp_circuit_data() {
        if (ctx) (void)hipSetDevice(ctx->device);
        for (void* q : dev) (void)hipFree(q);
    }
};

namespace {
template <typename T>
int dev_alloc(sipp_circuit_data* cd, size_t count, T** out) {
    void* q = nullptr;
    if (hipMalloc(&q, (count ? count : 1) * sizeof(T)) != hipSuccess) return sipp_fail(cd->ctx, SIPP_E_NOMEM, "circuit data: hipMalloc failed");
    cd->dev.push_back(q);
    *out = reinterpret_cast<T*>(q);
    return SIPP_OK;
}
template <typename T>
int dev_upload(sipp_circuit_data* cd, const T* host, size_t count, T** out) {
    SIPP_TRY(dev_alloc(cd, count, out));
    if (count) SIPP_CHECK_HIP(cd->ctx, hipMemcpy(*out, host, count * sizeof(T), hipMemcpyHostToDevice));
    return SIPP_OK;
}
uint32_t zs_columns(const sipp_plonk_params* p) { return p->num_challenges * ((p->num_routed_wires + p->max_degree - 1) / p->max_degree); }
}  // namespace

extern "C" size_t sipp_circuit_workspace_bytes(uint32_t log_n, const sipp_plonk_params* p, const sipp_fri_params* fp, const sipp_plonk_circuit* c) {
    if (!p || !fp || !c || log_n > 26 || !p->max_degree) return 0;
    // every oracle the prover commits inside the call (wires, Z / partial products, quotient chunks): coefficients + LDE, and the
    // quotient's own working set; the constants_sigmas oracle lives outside the arena
    const size_t n = (size_t)1 << log_n, cols = (size_t)c->num_wires + zs_columns(p) + (size_t)p->num_challenges * p->max_degree;
    return 8 * n * (((size_t)1 + ((size_t)1 << fp->rate_bits)) * (cols + c->num_constants + p->num_routed_wires) + 64) + ((size_t)4 << 30);
}

extern "C" void sipp_circuit_destroy(sipp_circuit_data* cd) {
    if (!cd) return;
    if (cd->ctx && cd->ctx->stream) {
        (void)hipSetDevice(cd->ctx->device);
        (void)hipStreamSynchronize(cd->ctx->stream);
        sipp_witness_graph_release(cd->ctx);          // a captured schedule names this circuit's buffers
    }
    delete cd;
}

extern "C" int sipp_circuit_build(sipp_ctx* ctx, uint32_t log_n, const sipp_plonk_params* p, const sipp_fri_params* fp, const sipp_plonk_circuit* c,
                                  const uint64_t* constants_sigmas, const sipp_plonk_generator* gens, size_t n_gens,
                                  const sipp_plonk_schedule_host* sched, const uint64_t* circuit_digest, sipp_circuit_data** out) {
    if (out) *out = nullptr;
    if (!ctx) return SIPP_E_BADARG;
    if (!out || !p || !fp || !c || !constants_sigmas || (!gens && n_gens) || log_n < 1 || log_n > 24 || !c->gates || (!c->programs && c->program_words) ||
        !c->num_gates || c->num_wires < p->num_routed_wires)
        return sipp_fail(ctx, SIPP_E_BADARG, "circuit build: null argument, log_n outside 1 .. 24 or an empty circuit");
    if (sched && sched->n_levels &&
        (!sched->rows || !sched->level_offsets || !sched->copy_offsets ||
         (sched->copy_offsets[sched->n_levels] && (!sched->copy_src || !sched->copy_dst)) || sched->n_levels > (1u << 20)))
        return sipp_fail(ctx, SIPP_E_BADARG, "circuit build: incomplete schedule");
    SIPP_TRY(sipp_plonk_circuit_check(ctx, c, p));          // a malformed gate set is refused here, not at the first proof
    SIPP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    sipp_circuit_data* cd = nullptr;
    try {
        cd = new sipp_circuit_data();
        cd->ctx = ctx; cd->log_n = log_n; cd->p = *p; cd->fp = *fp;
        cd->gates.assign(c->gates, c->gates + c->num_gates);
        cd->programs.assign(c->programs, c->programs + c->program_words);
        cd->gens.assign(gens, gens + n_gens);
        if (sched && sched->n_levels) {
            cd->level_offsets.assign(sched->level_offsets, sched->level_offsets + sched->n_levels + 1);
            cd->copy_offsets.assign(sched->copy_offsets, sched->copy_offsets + sched->n_levels + 1);
        }
    } catch (const std::bad_alloc&) {
        delete cd;
        return sipp_fail(ctx, SIPP_E_NOMEM, "circuit build: host allocation failed");
    }
    cd->circ = *c;
    cd->circ.gates = cd->gates.data();
    cd->circ.programs = cd->programs.data();
    auto bail = [&](int rc) {
        delete cd;
        return rc;
    };
    const size_t n = (size_t)1 << log_n, ncols = (size_t)c->num_constants + p->num_routed_wires, m = n << fp->rate_bits;
    int rc;
    if (sched && sched->n_levels) {
        const uint32_t L = sched->n_levels;
        for (uint32_t l = 0; l < L; l++)
            if (sched->level_offsets[l] > sched->level_offsets[l + 1] || sched->copy_offsets[l] > sched->copy_offsets[l + 1])
                return bail(sipp_fail(ctx, SIPP_E_BADARG, "circuit build: schedule offsets must not decrease"));
        if (sched->level_offsets[L] > n) return bail(sipp_fail(ctx, SIPP_E_BADARG, "circuit build: more scheduled rows than the table has"));
        uint32_t* d_rows = nullptr;
        uint64_t *d_src = nullptr, *d_dst = nullptr;
        if ((rc = dev_upload(cd, sched->rows, sched->level_offsets[L], &d_rows)) != SIPP_OK) return bail(rc);
        if ((rc = dev_upload(cd, sched->copy_src, sched->copy_offsets[L], &d_src)) != SIPP_OK) return bail(rc);
        if ((rc = dev_upload(cd, sched->copy_dst, sched->copy_offsets[L], &d_dst)) != SIPP_OK) return bail(rc);
        cd->has_sched = true;
        cd->sched.n_levels = L; cd->sched.d_rows = d_rows; cd->sched.d_copy_src = d_src; cd->sched.d_copy_dst = d_dst;
        cd->sched.level_offsets = cd->level_offsets.data(); cd->sched.copy_offsets = cd->copy_offsets.data();
    }
    uint64_t *d_coeffs = nullptr, *d_lde = nullptr, *d_tree = nullptr;
    if ((rc = dev_upload(cd, constants_sigmas, ncols * n, &cd->d_cs)) != SIPP_OK) return bail(rc);
    if ((rc = dev_alloc(cd, (size_t)c->num_wires * n, &cd->d_wires)) != SIPP_OK) return bail(rc);
    if ((rc = dev_alloc(cd, ncols * n, &d_coeffs)) != SIPP_OK) return bail(rc);
    if ((rc = dev_alloc(cd, ncols * m, &d_lde)) != SIPP_OK) return bail(rc);
    if ((rc = dev_alloc(cd, 2 * m * 4, &d_tree)) != SIPP_OK) return bail(rc);
    const uint32_t ch = fp->cap_height < log_n + fp->rate_bits ? fp->cap_height : log_n + fp->rate_bits;
    cd->cap.assign(((size_t)4) << ch, 0);
    rc = sipp_commit_batch_ex(ctx, cd->d_cs, 0, d_coeffs, d_lde, d_tree, ncols, log_n, fp->rate_bits, fp->cap_height, nullptr, 0, cd->cap.data());
    if (rc != SIPP_OK) return bail(rc);
    SIPP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    cd->cs_oracle.d_coeffs = d_coeffs; cd->cs_oracle.d_lde = d_lde; cd->cs_oracle.d_tree = d_tree;
    cd->cs_oracle.n_polys = (uint32_t)ncols; cd->cs_oracle.n_salt = 0;
    if (circuit_digest) {
        for (int q = 0; q < 4; q++) cd->digest[q] = circuit_digest[q];
    } else {
        std::vector<uint64_t> w(cd->cap);
        for (uint64_t v : {(uint64_t)log_n, (uint64_t)c->num_wires, (uint64_t)p->num_routed_wires, (uint64_t)c->num_constants, (uint64_t)c->num_selectors,
                           (uint64_t)c->num_gates})
            w.push_back(v);
        host::Challenger::hash_no_pad(w.data(), w.size(), cd->digest);
    }
    if (sipp_plonk_gates_proof_size(log_n, p, fp, &cd->circ, 0) == 0) return bail(sipp_fail(ctx, SIPP_E_BADARG, "circuit build: parameters or gate set refused"));
    *out = cd;
    return SIPP_OK;
}

extern "C" int sipp_circuit_verifier_data(const sipp_circuit_data* cd, uint64_t* cap_out, uint64_t* digest_out) {
    if (!cd || !cap_out || !digest_out) return SIPP_E_BADARG;
    std::copy(cd->cap.begin(), cd->cap.end(), cap_out);
    for (int q = 0; q < 4; q++) digest_out[q] = cd->digest[q];
    return SIPP_OK;
}

extern "C" size_t sipp_circuit_proof_size(const sipp_circuit_data* cd, uint32_t n_public_inputs) {
    return cd ? sipp_plonk_gates_proof_size(cd->log_n, &cd->p, &cd->fp, &cd->circ, n_public_inputs) : 0;
}

extern "C" int sipp_circuit_prove(sipp_circuit_data* cd, const uint64_t* wires, const uint64_t* public_inputs, uint32_t n_public_inputs,
                                  uint64_t* proof_out, size_t proof_cap, size_t* proof_len) {
    if (!cd) return SIPP_E_BADARG;
    sipp_ctx* ctx = cd->ctx;
    if (!wires || !proof_out || !proof_len || (n_public_inputs && !public_inputs) || n_public_inputs > (1u << 24))
        return sipp_fail(ctx, SIPP_E_BADARG, "circuit prove: null argument");
    SIPP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)1 << cd->log_n;
    SIPP_CHECK_HIP(ctx, hipMemcpyAsync(cd->d_wires, wires, (size_t)cd->circ.num_wires * n * 8, hipMemcpyHostToDevice, ctx->stream));
    uint64_t pih[4];
    host::Challenger::hash_no_pad(public_inputs, n_public_inputs, pih);
    if (cd->has_sched)
        SIPP_TRY(sipp_plonk_generate_witness_levels(ctx, cd->d_wires, cd->d_cs, cd->log_n, cd->circ.num_wires, cd->circ.num_constants, cd->gens.data(),
                                                    cd->gens.size(), pih, &cd->sched));
    else
        SIPP_TRY(sipp_plonk_generate_witness(ctx, cd->d_wires, cd->d_cs, cd->log_n, cd->circ.num_wires, cd->circ.num_constants, cd->gens.data(),
                                             cd->gens.size(), pih));
    return sipp_plonk_prove_gates(ctx, cd->d_wires, cd->d_cs, nullptr, nullptr, &cd->cs_oracle, cd->log_n, &cd->p, &cd->fp, &cd->circ, cd->digest,
                                  public_inputs, n_public_inputs, proof_out, proof_cap, proof_len);
}

extern "C" int sipp_circuit_verify(const sipp_circuit_data* cd, const uint64_t* proof, size_t len, int* reason) {
    if (!cd) return SIPP_E_BADARG;
    return sipp_plonk_verify_gates(proof, len, cd->cap.data(), &cd->p, &cd->fp, &cd->circ, cd->digest, reason);
}
