// sipp_host.hpp -- C++ host side above the C ABI of sipp_hip.h.
//
// The reference is Rust (no toolchain in this image), so the host layer a maintainer would write in Rust is written
// here in C++ with the reference's own names and argument meaning:
//
//   reference (src/verifier_circuit.rs)                              here
//   -------------------------------------------------------------    ------------------------------------------------
//   G1ExpInputTarget  { x, offset, exp_val }            :92-96       sipp::G1ExpInput   { x, offset, exp_val }
//   G2ExpInputTarget  { x, offset, exp_val }            :101-105     sipp::G2ExpInput
//   Fq12ExpInputTarget{ x, offset, exp_val }            :111-115     sipp::Fq12ExpInput
//   g1_exp_circuit(builder, &inputs)  -> Vec<G1Target>  :133         sipp::Prover::g1_exp_circuit(inputs)  -> outputs + proof
//   g2_exp_circuit / fq12_exp_circuit                   :134-135     sipp::Prover::g2_exp_circuit / fq12_exp_circuit
//   StarkProofWithPublicInputs<F, C, D> (starky)                     sipp::StarkProofWithPublicInputs (same field names)
//   batch_map_to_g2_circuit(builder, &messages)  (src/bin/bls_aggregation.rs:65)   sipp::Prover::batch_map_to_g2_circuit(messages)
//   map_to_g2_without_cofactor_mul(u).mul_by_cofactor()     (:100-104)            sipp::Prover::map_to_g2(messages)
//   anyhow::Error at the .unwrap() of :253                           sipp::Error (carries the sipp_status)
//
// In the reference the three calls add a recursive verifier gadget and register a witness generator; at proving time the
// generator computes the outputs natively, fills the trace and runs starky::prover::prove.  Prover::*_exp_circuit is that
// proving-time body: outputs (sipp_exp_outputs) and proof (sipp_*_exp_prove) both come from the GPU.
// Header-only; link with -lsipp_hip.  No torch, no HIP types.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "sipp_hip.h"

namespace sipp {

// ---- BN254 values as the reference serialises them: 8 x u32 little-endian limbs (transcript_native.rs:68-77) ----
using U256 = std::array<uint32_t, 8>;
struct Fq2 {
    U256 c0, c1;  // statements.rs:102-119 order
};
struct G1Affine {
    U256 x, y;
};
struct G2Affine {
    Fq2 x, y;
};
struct Fq12 {
    std::array<U256, 12> c;  // MyFq12 coefficient order (verifier_circuit.rs:111-124)
};

struct G1ExpInput {
    G1Affine x, offset;
    U256 exp_val;
};
struct G2ExpInput {
    G2Affine x, offset;
    U256 exp_val;
};
struct Fq12ExpInput {
    Fq12 x, offset;
    U256 exp_val;
};

// one IO record of the C ABI = input followed by the output
template <class In, class Out>
struct ExpIO {
    In in;
    Out out;
};
using G1ExpIO = ExpIO<G1ExpInput, G1Affine>;
using G2ExpIO = ExpIO<G2ExpInput, G2Affine>;
using Fq12ExpIO = ExpIO<Fq12ExpInput, Fq12>;
static_assert(sizeof(G1ExpIO) == 4 * SIPP_G1_IO_WORDS, "G1 IO record layout");
static_assert(sizeof(G2ExpIO) == 4 * SIPP_G2_IO_WORDS, "G2 IO record layout");
static_assert(sizeof(Fq12ExpIO) == 4 * SIPP_FQ12_IO_WORDS, "Fq12 IO record layout");
// MapToG2 record: the message u in Fp2, then its point on the twist
using MapG2IO = ExpIO<Fq2, G2Affine>;
static_assert(sizeof(MapG2IO) == 4 * SIPP_MAP_G2_IO_WORDS, "MapToG2 IO record layout");
// final-pairing record: (P, Q), then e(P, Q)
struct PairingInput {
    G1Affine p;
    G2Affine q;
};
using PairingIO = ExpIO<PairingInput, Fq12>;
static_assert(sizeof(PairingIO) == 4 * SIPP_PAIRING_IO_WORDS, "pairing IO record layout");

// ---- errors: the reference's call sites unwrap an anyhow::Result; here a failing status throws ----
class Error : public std::runtime_error {
   public:
    Error(int status, const std::string& what) : std::runtime_error(what), status_(status) {}
    int status() const { return status_; }

   private:
    int status_;
};

// starky's `verify_stark_proof` as the generators call it after `prove`, and the check `data.verify(proof)` rests on (reference
// src/verifier_circuit.rs:254): the library's own verifier (sipp_stark_verify: host code, no GPU).  Throws Error(SIPP_E_VERIFY)
// naming the refusing stage (include/sipp_hip.h).
inline void verify_stark_proof(const std::vector<uint64_t>& flat, const sipp_stark_config* cfg = nullptr) {
    int reason = 0;
    const int rc = ::sipp_stark_verify(flat.data(), flat.size(), cfg, &reason);
    if (rc != SIPP_OK) throw Error(rc, "verify_stark_proof: refused at stage " + std::to_string(reason));
}

// `data.verify(proof)` for the OUTER proof (reference src/verifier_circuit.rs:254; plonk/verifier.rs) of sipp_plonk_prove_gates: verifier data =
// the constants_sigmas cap and the circuit digest; the gate set is the data the prover interpreted.  Throws Error(SIPP_E_VERIFY) naming the stage.
inline void verify_plonk_proof(const std::vector<uint64_t>& flat, const std::vector<uint64_t>& constants_sigmas_cap, const sipp_plonk_params& params,
                               const sipp_fri_params& fri, const sipp_plonk_circuit& circuit, const uint64_t circuit_digest[4]) {
    int reason = 0;
    const int rc = ::sipp_plonk_verify_gates(flat.data(), flat.size(), constants_sigmas_cap.data(), &params, &fri, &circuit, circuit_digest, &reason);
    if (rc != SIPP_OK) throw Error(rc, "verify_plonk_proof: refused at stage " + std::to_string(reason));
}

// ---- starky's proof structs (field names of starky::proof / plonky2::fri::proof) ----
using F = uint64_t;  // canonical Goldilocks
struct Ext {
    F c0, c1;
    bool operator==(const Ext& o) const { return c0 == o.c0 && c1 == o.c1; }
};
using HashOut = std::array<F, 4>;
using MerkleCap = std::vector<HashOut>;
struct MerkleProof {
    std::vector<HashOut> siblings;
};
struct StarkOpeningSet {
    std::vector<Ext> local_values, next_values, permutation_zs, permutation_zs_next, quotient_polys;
};
struct FriInitialTreeProof {
    std::vector<std::pair<std::vector<F>, MerkleProof>> evals_proofs;  // trace, permutation Zs, quotient
};
struct FriQueryStep {
    std::vector<Ext> evals;
    MerkleProof merkle_proof;
};
struct FriQueryRound {
    FriInitialTreeProof initial_trees_proof;
    std::vector<FriQueryStep> steps;
};
struct FriProof {
    std::vector<MerkleCap> commit_phase_merkle_caps;
    std::vector<FriQueryRound> query_round_proofs;
    std::vector<Ext> final_poly;
    F pow_witness = 0;
};
struct StarkProof {
    MerkleCap trace_cap, permutation_zs_cap, quotient_polys_cap;
    StarkOpeningSet openings;
    FriProof opening_proof;
};

struct StarkProofWithPublicInputs {
    // shape (the flat buffer's header)
    uint32_t kind = 0, degree_bits = 0, num_io = 0, main_cols = 0, perm_cols = 0, quotient_cols = 0, cap_height = 0,
             pi_per_io = 0, rate_bits = 1, arity_bits = 4;
    StarkProof proof;
    std::vector<F> public_inputs;  // the (padded) IO records, one field element per u32 word

    static constexpr uint64_t MAGIC = 0x5349505053544b31ULL;  // "SIPPSTK1"

    static StarkProofWithPublicInputs from_flat(const uint64_t* w, size_t len) {
        if (len < 16 || w[0] != MAGIC || w[12] != len) throw Error(SIPP_E_BADARG, "from_flat: not a SIPP STARK proof buffer");
        // the header of a buffer from anywhere is untrusted: every field that sizes an allocation or a shift is bounded first
        // (each of the counts below is also at most `len`, or the section checks would have to refuse the buffer anyway)
        for (int i = 1; i < 16; i++)
            if (w[i] >> 32) throw Error(SIPP_E_BADARG, "from_flat: header field out of range");
        if (w[2] > 30 || w[7] > 16 || w[13] < 1 || w[13] > 3 || w[14] < 1 || w[14] > 4 || w[15] != 0 || w[8] > 32 || w[9] > len ||
            w[10] > len || w[4] > len || w[5] > len || w[6] > len)
            throw Error(SIPP_E_BADARG, "from_flat: header field out of range");
        for (size_t i = 16; i < len; i++)   // canonical field elements only: x + p would be a second encoding of x
            if (w[i] >= 0xFFFFFFFF00000001ULL) throw Error(SIPP_E_BADARG, "from_flat: non-canonical field element");
        StarkProofWithPublicInputs p;
        p.kind = (uint32_t)w[1];
        p.degree_bits = (uint32_t)w[2];
        p.num_io = (uint32_t)w[3];
        p.main_cols = (uint32_t)w[4];
        p.perm_cols = (uint32_t)w[5];
        p.quotient_cols = (uint32_t)w[6];
        p.cap_height = (uint32_t)w[7];
        p.pi_per_io = (uint32_t)w[11];
        p.rate_bits = (uint32_t)w[13];
        p.arity_bits = (uint32_t)w[14];
        const uint32_t rounds = (uint32_t)w[8], final_len = (uint32_t)w[9], nq = (uint32_t)w[10];
        const uint32_t log_m = p.degree_bits + p.rate_bits, ncap = 1u << p.cap_height;
        if (p.cap_height > log_m) throw Error(SIPP_E_BADARG, "from_flat: cap higher than the tree");
        size_t pos = 16;
        auto need = [&](size_t cnt) {
            if (cnt > len - pos) throw Error(SIPP_E_BUFSZ, "from_flat: truncated proof");
        };
        auto hashes = [&](size_t cnt) {
            need(4 * cnt);
            std::vector<HashOut> v(cnt);
            for (size_t i = 0; i < cnt; i++, pos += 4) v[i] = HashOut{w[pos], w[pos + 1], w[pos + 2], w[pos + 3]};
            return v;
        };
        auto exts = [&](size_t cnt) {
            need(2 * cnt);
            std::vector<Ext> v(cnt);
            for (size_t i = 0; i < cnt; i++, pos += 2) v[i] = Ext{w[pos], w[pos + 1]};
            return v;
        };
        auto bases = [&](size_t cnt) {
            need(cnt);
            std::vector<F> v(w + pos, w + pos + cnt);
            pos += cnt;
            return v;
        };
        StarkProof& s = p.proof;
        s.trace_cap = hashes(ncap);
        s.permutation_zs_cap = hashes(ncap);
        s.quotient_polys_cap = hashes(ncap);
        s.openings.local_values = exts(p.main_cols);
        s.openings.next_values = exts(p.main_cols);
        s.openings.permutation_zs = exts(p.perm_cols);
        s.openings.permutation_zs_next = exts(p.perm_cols);
        s.openings.quotient_polys = exts(p.quotient_cols);
        FriProof& f = s.opening_proof;
        for (uint32_t r = 0; r < rounds; r++) {
            if (p.arity_bits * (r + 1) > log_m) throw Error(SIPP_E_BADARG, "from_flat: more FRI rounds than the domain has bits");
            const uint32_t lt = log_m - p.arity_bits * (r + 1);          // leaves of round r's tree (log2)
            f.commit_phase_merkle_caps.push_back(hashes((size_t)1 << (lt < p.cap_height ? lt : p.cap_height)));
        }
        f.final_poly = exts(final_len);
        need(1);
        f.pow_witness = w[pos++];
        const uint32_t cols[3] = {p.main_cols, p.perm_cols, p.quotient_cols};
        f.query_round_proofs.resize(nq);
        for (uint32_t q = 0; q < nq; q++) {
            FriQueryRound& qr = f.query_round_proofs[q];
            for (int o = 0; o < 3; o++) {
                std::vector<F> row = bases(cols[o]);
                qr.initial_trees_proof.evals_proofs.emplace_back(std::move(row), MerkleProof{hashes(log_m - p.cap_height)});
            }
            for (uint32_t r = 0; r < rounds; r++) {
                if (p.arity_bits * (r + 1) > log_m) throw Error(SIPP_E_BADARG, "from_flat: more FRI rounds than the domain has bits");
                const uint32_t lt = log_m - p.arity_bits * (r + 1);      // leaves of round r's tree (log2)
                FriQueryStep st;
                st.evals = exts((size_t)1 << p.arity_bits);
                st.merkle_proof.siblings = hashes(lt > p.cap_height ? lt - p.cap_height : 0);
                qr.steps.push_back(std::move(st));
            }
        }
        if ((uint64_t)p.num_io * p.pi_per_io > len) throw Error(SIPP_E_BUFSZ, "from_flat: truncated proof");
        p.public_inputs = bases((size_t)p.num_io * p.pi_per_io);
        if (pos != len) throw Error(SIPP_E_BUFSZ, "from_flat: trailing words");
        return p;
    }

    std::vector<uint64_t> to_flat() const {
        std::vector<uint64_t> w(16, 0);
        const StarkProof& s = proof;
        const FriProof& f = s.opening_proof;
        auto hashes = [&](const std::vector<HashOut>& v) {
            for (const HashOut& h : v) w.insert(w.end(), h.begin(), h.end());
        };
        auto exts = [&](const std::vector<Ext>& v) {
            for (const Ext& e : v) {
                w.push_back(e.c0);
                w.push_back(e.c1);
            }
        };
        hashes(s.trace_cap);
        hashes(s.permutation_zs_cap);
        hashes(s.quotient_polys_cap);
        exts(s.openings.local_values);
        exts(s.openings.next_values);
        exts(s.openings.permutation_zs);
        exts(s.openings.permutation_zs_next);
        exts(s.openings.quotient_polys);
        for (const MerkleCap& c : f.commit_phase_merkle_caps) hashes(c);
        exts(f.final_poly);
        w.push_back(f.pow_witness);
        for (const FriQueryRound& qr : f.query_round_proofs) {
            for (const auto& ep : qr.initial_trees_proof.evals_proofs) {
                w.insert(w.end(), ep.first.begin(), ep.first.end());
                hashes(ep.second.siblings);
            }
            for (const FriQueryStep& st : qr.steps) {
                exts(st.evals);
                hashes(st.merkle_proof.siblings);
            }
        }
        w.insert(w.end(), public_inputs.begin(), public_inputs.end());
        const uint64_t hdr[15] = {MAGIC, kind, degree_bits, num_io, main_cols, perm_cols, quotient_cols, cap_height,
                                  (uint64_t)f.commit_phase_merkle_caps.size(), (uint64_t)f.final_poly.size(),
                                  (uint64_t)f.query_round_proofs.size(), pi_per_io, (uint64_t)w.size(), rate_bits, arity_bits};
        std::memcpy(w.data(), hdr, sizeof hdr);
        return w;
    }
};

// ---- the OUTER proof: plonky2's CircuitData (reference src/verifier_circuit.rs:225 `builder.build::<C>()`, :253 `data.prove(pw)`,
// :254 `data.verify(proof)`) over sipp_circuit_build / _prove / _verify.  The circuit -- gate set, generators, level schedule,
// constants_sigmas values -- is DATA the caller's circuit builder produced (the reference's own builder lives in un-vendored crates);
// everything stays in host memory on this side.
struct ProofWithPublicInputs {
    std::vector<uint64_t> flat;            // "SIPPPLK3": header | caps | opening proof | public inputs
    std::vector<uint64_t> public_inputs;   // the tail of `flat` (SIPPStatement::from_vec reads these back: verifier_circuit.rs:258-268)
};
class CircuitData {
   public:
    // degree_bits = log2 rows; constants_sigmas [num_constants + num_routed_wires][N] values; schedule may be null (row-local generators);
    // digest may be null (derived from the constants_sigmas cap and the shape)
    CircuitData(int device, uint32_t degree_bits, const sipp_plonk_params& params, const sipp_fri_params& fri, const sipp_plonk_circuit& circuit,
                const std::vector<uint64_t>& constants_sigmas, const std::vector<sipp_plonk_generator>& generators,
                const sipp_plonk_schedule_host* schedule = nullptr, const uint64_t* digest = nullptr)
        : num_wires_(circuit.num_wires), n_((size_t)1 << degree_bits) {
        if (constants_sigmas.size() != ((size_t)circuit.num_constants + params.num_routed_wires) * n_)
            throw Error(SIPP_E_BADARG, "CircuitData: constants_sigmas must be [num_constants + num_routed_wires][N]");
        int rc = sipp_ctx_create(&ctx_, device, nullptr, sipp_circuit_workspace_bytes(degree_bits, &params, &fri, &circuit));
        if (rc != SIPP_OK) throw Error(rc, "CircuitData: sipp_ctx_create failed");
        rc = sipp_circuit_build(ctx_, degree_bits, &params, &fri, &circuit, constants_sigmas.data(), generators.data(), generators.size(), schedule,
                                digest, &data_);
        if (rc != SIPP_OK) {
            const std::string msg = std::string("sipp_circuit_build: ") + sipp_last_error(ctx_);
            sipp_ctx_destroy(ctx_);
            throw Error(rc, msg);
        }
        constants_sigmas_cap.assign((size_t)4 << std::min(fri.cap_height, degree_bits + fri.rate_bits), 0);
        (void)sipp_circuit_verifier_data(data_, constants_sigmas_cap.data(), circuit_digest);
    }
    ~CircuitData() {
        sipp_circuit_destroy(data_);
        sipp_ctx_destroy(ctx_);
    }
    CircuitData(const CircuitData&) = delete;
    CircuitData& operator=(const CircuitData&) = delete;

    // `data.prove(pw)`: pw = the wire table [num_wires][N] with the input cells set (the generated cells are overwritten on the device)
    ProofWithPublicInputs prove(const std::vector<uint64_t>& partial_witness, const std::vector<uint64_t>& public_inputs) {
        if (partial_witness.size() != (size_t)num_wires_ * n_) throw Error(SIPP_E_BADARG, "CircuitData::prove: the witness must be [num_wires][N]");
        ProofWithPublicInputs r;
        const size_t cap = sipp_circuit_proof_size(data_, (uint32_t)public_inputs.size());
        r.flat.assign(cap, 0);
        size_t len = 0;
        const int rc = sipp_circuit_prove(data_, partial_witness.data(), public_inputs.data(), (uint32_t)public_inputs.size(), r.flat.data(), cap, &len);
        if (rc != SIPP_OK) throw Error(rc, std::string("sipp_circuit_prove: ") + sipp_last_error(ctx_));
        r.flat.resize(len);
        r.public_inputs.assign(r.flat.end() - (std::ptrdiff_t)public_inputs.size(), r.flat.end());
        return r;
    }
    // `data.verify(proof)`: throws Error(SIPP_E_VERIFY) naming the refusing stage
    void verify(const ProofWithPublicInputs& proof) const {
        int reason = 0;
        const int rc = sipp_circuit_verify(data_, proof.flat.data(), proof.flat.size(), &reason);
        if (rc != SIPP_OK) throw Error(rc, "CircuitData::verify: refused at stage " + std::to_string(reason));
    }
    std::vector<uint64_t> constants_sigmas_cap;   // VerifierOnlyCircuitData
    uint64_t circuit_digest[4] = {0, 0, 0, 0};

   private:
    sipp_ctx* ctx_ = nullptr;
    sipp_circuit_data* data_ = nullptr;
    uint32_t num_wires_;
    size_t n_;
};

template <class Out>
struct ExpCircuitResult {
    std::vector<Out> outputs;          // what g*_exp_circuit returns as targets, as values
    StarkProofWithPublicInputs proof;  // what the generator writes into the recursive verifier's proof target
    std::vector<uint64_t> flat;        // the same proof as the C ABI returned it
};

// One GPU, three sub-provers (three ctxs = three HIP streams), sized for the IO counts given at construction
// (no device allocation afterwards, SURVEY.md section 8b "threading").
class Prover {
   public:
    Prover(int device, size_t max_g1_io, size_t max_g2_io, size_t max_fq12_io) {
        const size_t mx[3] = {max_g1_io, max_g2_io, max_fq12_io};
        for (int k = 0; k < 3; k++) {
            max_io_[k] = mx[k];
            // (the G1 / G2 arenas also hold the hardened AIRs of sipp_hip.h kinds 4 / 5: about 14 % more columns, 11 % more time)
            size_t ws = k < 2 ? sipp_workspace_bytes(k + SIPP_G1_EXP_HARDENED, mx[k]) : sipp_workspace_bytes(k, mx[k]);
            if (k == SIPP_G2_EXP) ws = std::max(ws, sipp_workspace_bytes(SIPP_PAIRING, 1));   // pairing_circuit runs there
            const int rc = sipp_ctx_create(&ctx_[k], device, nullptr, ws);
            if (rc != SIPP_OK) {
                for (int j = 0; j < k; j++) sipp_ctx_destroy(ctx_[j]);
                throw Error(rc, "sipp_ctx_create failed (see stderr)");
            }
            static const int level[3] = {-1, 0, 1};  // G1 low, G2 normal, Fq12 high
            (void)sipp_ctx_set_stream_priority(ctx_[k], level[k]);
        }
    }
    ~Prover() {
        for (sipp_ctx* c : ctx_) sipp_ctx_destroy(c);
    }
    Prover(const Prover&) = delete;
    Prover& operator=(const Prover&) = delete;
    // every *_circuit call verifies its proof before returning it (default: on, as upstream's generators do)
    void set_verify_after_prove(bool on) { verify_after_prove_ = on; }

    // hardened = true: the same obligations proved with the hardened curve AIR (sipp_hip.h kinds 4 / 5): for
    // statements whose offsets an adversary may choose
    ExpCircuitResult<G1Affine> g1_exp_circuit(const std::vector<G1ExpInput>& inputs, bool hardened = false) {
        return run<G1ExpInput, G1Affine>(SIPP_G1_EXP, inputs, hardened);
    }
    ExpCircuitResult<G2Affine> g2_exp_circuit(const std::vector<G2ExpInput>& inputs, bool hardened = false) {
        return run<G2ExpInput, G2Affine>(SIPP_G2_EXP, inputs, hardened);
    }
    ExpCircuitResult<Fq12> fq12_exp_circuit(const std::vector<Fq12ExpInput>& inputs) {
        return run<Fq12ExpInput, Fq12>(SIPP_FQ12_EXP, inputs);
    }

    // batch_map_to_g2_circuit(builder, &messages) of the BLS example (src/bin/bls_aggregation.rs:65), proving-time body: the points
    // map_to_g2_without_cofactor_mul(u) (outputs) and the MapToG2 proof over the records (u, x, y).  Runs on the G2 ctx (its arena
    // is the larger one whenever max_g2_io >= messages / 64).  The cofactor multiplication that follows in the reference is a pair
    // of g2_exp_circuit obligations per message: map_to_g2() below returns them.
    ExpCircuitResult<G2Affine> batch_map_to_g2_circuit(const std::vector<Fq2>& messages) {
        if (messages.empty()) throw Error(SIPP_E_BADARG, "batch_map_to_g2_circuit: no messages");
        sipp_ctx* c = ctx_[SIPP_G2_EXP];
        std::vector<MapG2IO> io(messages.size());
        for (size_t i = 0; i < io.size(); i++) {
            io[i].in = messages[i];
            std::memset(&io[i].out, 0, sizeof(G2Affine));
        }
        int rc = sipp_exp_outputs(c, SIPP_MAP_G2, reinterpret_cast<uint32_t*>(io.data()), io.size());
        if (rc != SIPP_OK) throw Error(rc, std::string("sipp_exp_outputs(MAP_G2): ") + sipp_last_error(c));
        ExpCircuitResult<G2Affine> r;
        const size_t cap = sipp_proof_size(c, SIPP_MAP_G2, io.size());
        r.flat.assign(cap, 0);
        size_t len = 0;
        rc = sipp_map_to_g2_prove(c, words(io), io.size(), r.flat.data(), cap, &len);
        if (rc != SIPP_OK) throw Error(rc, std::string("sipp_map_to_g2_prove: ") + sipp_last_error(c));
        r.flat.resize(len);
        finish(io, &r);
        return r;
    }
    // pairing_circuit(builder, final_A, final_B) of the BLS example (src/bin/bls_aggregation.rs:76), proving-time body: the value
    // e(final_A, final_B) (output; the caller connects it to final_Z, :77) and the final-pairing proof over the record (P, Q, Z).
    // Several pairs give one proof over several records.  Runs on the G2 ctx.
    ExpCircuitResult<Fq12> pairing_circuit(const std::vector<PairingInput>& pairs) {
        if (pairs.empty()) throw Error(SIPP_E_BADARG, "pairing_circuit: no pairs");
        sipp_ctx* c = ctx_[SIPP_G2_EXP];
        std::vector<PairingIO> io(pairs.size());
        for (size_t i = 0; i < io.size(); i++) {
            io[i].in = pairs[i];
            std::memset(&io[i].out, 0, sizeof(Fq12));
        }
        int rc = sipp_exp_outputs(c, SIPP_PAIRING, reinterpret_cast<uint32_t*>(io.data()), io.size());
        if (rc != SIPP_OK) throw Error(rc, std::string("sipp_exp_outputs(PAIRING): ") + sipp_last_error(c));
        ExpCircuitResult<Fq12> r;
        const size_t cap = sipp_proof_size(c, SIPP_PAIRING, io.size());
        r.flat.assign(cap, 0);
        size_t len = 0;
        rc = sipp_pairing_prove(c, words(io), io.size(), r.flat.data(), cap, &len);
        if (rc != SIPP_OK) throw Error(rc, std::string("sipp_pairing_prove: ") + sipp_last_error(c));
        r.flat.resize(len);
        finish(io, &r);
        return r;
    }
    ExpCircuitResult<Fq12> pairing_circuit(const G1Affine& final_A, const G2Affine& final_B) {
        return pairing_circuit(std::vector<PairingInput>{PairingInput{final_A, final_B}});
    }
    // messages.iter().map(|u| map_to_g2_without_cofactor_mul(*u).mul_by_cofactor()) (src/bin/bls_aggregation.rs:100-104): the points
    // of G2, and the 2 n g2_exp_circuit inputs whose outputs clear the cofactor (G + [2p - r] Q, then - G)
    struct MappedMessages {
        std::vector<G2Affine> points;            // `ms`
        std::vector<G2ExpInput> cofactor_inputs; // obligations for g2_exp_circuit
    };
    MappedMessages map_to_g2(const std::vector<Fq2>& messages) {
        if (messages.empty()) throw Error(SIPP_E_BADARG, "map_to_g2: no messages");
        sipp_ctx* c = ctx_[SIPP_G2_EXP];
        const size_t n = messages.size();
        std::vector<MapG2IO> io(n);
        std::vector<G2ExpIO> g2(2 * n);
        MappedMessages m;
        m.points.resize(n);
        const int rc = sipp_map_to_g2(c, reinterpret_cast<const uint32_t*>(messages.data()), n, reinterpret_cast<uint32_t*>(io.data()),
                                      reinterpret_cast<uint32_t*>(g2.data()), reinterpret_cast<uint32_t*>(m.points.data()));
        if (rc != SIPP_OK) throw Error(rc, std::string("sipp_map_to_g2: ") + sipp_last_error(c));
        m.cofactor_inputs.resize(2 * n);
        for (size_t i = 0; i < 2 * n; i++) m.cofactor_inputs[i] = g2[i].in;
        return m;
    }

    // the three calls of verifier_circuit.rs:133-135 together: outputs first, then the three proofs concurrently
    void exp_circuits(const std::vector<G1ExpInput>& g1, const std::vector<G2ExpInput>& g2, const std::vector<Fq12ExpInput>& fq12,
                      ExpCircuitResult<G1Affine>* r1, ExpCircuitResult<G2Affine>* r2, ExpCircuitResult<Fq12>* r12) {
        std::vector<G1ExpIO> io1 = outputs<G1ExpInput, G1Affine>(SIPP_G1_EXP, g1);
        std::vector<G2ExpIO> io2 = outputs<G2ExpInput, G2Affine>(SIPP_G2_EXP, g2);
        std::vector<Fq12ExpIO> io12 = outputs<Fq12ExpInput, Fq12>(SIPP_FQ12_EXP, fq12);
        const uint32_t* ios[3] = {words(io1), words(io2), words(io12)};
        const size_t num[3] = {io1.size(), io2.size(), io12.size()};
        size_t cap[3], len[3] = {0, 0, 0};
        std::vector<uint64_t>* flat[3] = {&r1->flat, &r2->flat, &r12->flat};
        uint64_t* out[3];
        for (int k = 0; k < 3; k++) {
            cap[k] = sipp_proof_size(ctx_[k], k, num[k]);
            flat[k]->assign(cap[k], 0);
            out[k] = flat[k]->data();
        }
        const int rc = sipp_instance_prove(ctx_, ios, num, out, cap, len);
        if (rc != SIPP_OK) throw Error(rc, std::string("sipp_instance_prove: ") + first_error());
        for (int k = 0; k < 3; k++) flat[k]->resize(len[k]);
        finish(io1, r1);
        finish(io2, r2);
        finish(io12, r12);
    }

    // Several GPUs for ONE instance without any exchange (level L-D): the slice of an obligation list that
    // GPU `rank` of `world` hands to its own g1_exp_circuit / g2_exp_circuit / fq12_exp_circuit call (sipp_io_shard)
    template <class In>
    static std::vector<In> io_shard(const std::vector<In>& all, unsigned world, unsigned rank) {
        size_t first = 0, count = 0;
        const int rc = sipp_io_shard(all.size(), world, rank, &first, &count);
        if (rc != SIPP_OK) throw Error(rc, "sipp_io_shard: rank >= world or world == 0");
        return std::vector<In>(all.begin() + (ptrdiff_t)first, all.begin() + (ptrdiff_t)(first + count));
    }

    // ---- the native chain in front of the circuit ----
    // reference src/prover_native.rs:26-80: the 2 log2 n + 1 proof messages, in the reference's (reversed) order
    std::vector<Fq12> sipp_prove_native(const std::vector<G1Affine>& A, const std::vector<G2Affine>& B) {
        if (A.size() != B.size() || sipp_native_proof_words(A.size()) == 0) throw Error(SIPP_E_BADARG, "sipp_prove_native: |A| = |B| = 2^k");
        std::vector<Fq12> proof(sipp_native_proof_words(A.size()) / 96);
        const int rc = ::sipp_prove_native(ctx_[SIPP_G2_EXP], reinterpret_cast<const uint32_t*>(A.data()),
                                           reinterpret_cast<const uint32_t*>(B.data()), A.size(), reinterpret_cast<uint32_t*>(proof.data()));
        if (rc != SIPP_OK) throw Error(rc, std::string("sipp_prove_native: ") + sipp_last_error(ctx_[SIPP_G2_EXP]));
        return proof;
    }
    // reference src/statements.rs:14-22
    struct SIPPStatement {
        std::vector<G1Affine> A;
        std::vector<G2Affine> B;
        Fq12 Z;
        G1Affine final_A;
        G2Affine final_B;
        Fq12 final_Z;
    };
    struct NativeVerification {
        SIPPStatement statement;
        std::vector<G1ExpIO> g1_obligations;      // verifier_circuit.rs:92-98, outputs included
        std::vector<G2ExpIO> g2_obligations;      // :101-107
        std::vector<Fq12ExpIO> fq12_obligations;  // :111-124
    };
    // reference src/verifier_native.rs:14-85; throws Error(SIPP_E_WITNESS) where the reference returns Err("Verification failed")
    NativeVerification sipp_verify_native(const std::vector<G1Affine>& A, const std::vector<G2Affine>& B, const std::vector<Fq12>& proof) {
        const size_t n = A.size();
        if (n != B.size() || proof.size() * 96 != sipp_native_proof_words(n)) throw Error(SIPP_E_BADARG, "sipp_verify_native: sizes");
        size_t lg = 0;
        while (((size_t)1 << lg) < n) lg++;
        NativeVerification v;
        v.g1_obligations.resize(n - 1);
        v.g2_obligations.resize(n - 1);
        v.fq12_obligations.resize(2 * lg);
        std::vector<uint32_t> st(48 * n + 240);
        int ok = 0;
        const int rc = ::sipp_verify_native(ctx_[SIPP_G2_EXP], reinterpret_cast<const uint32_t*>(A.data()),
                                            reinterpret_cast<const uint32_t*>(B.data()), n, reinterpret_cast<const uint32_t*>(proof.data()),
                                            st.data(), reinterpret_cast<uint32_t*>(v.g1_obligations.data()),
                                            reinterpret_cast<uint32_t*>(v.g2_obligations.data()),
                                            reinterpret_cast<uint32_t*>(v.fq12_obligations.data()), &ok);
        if (rc != SIPP_OK) throw Error(rc, std::string("sipp_verify_native: ") + sipp_last_error(ctx_[SIPP_G2_EXP]));
        if (!ok) throw Error(SIPP_E_WITNESS, "Verification failed");
        v.statement.A = A;
        v.statement.B = B;
        const uint32_t* s = st.data() + 48 * n;
        std::memcpy(&v.statement.Z, s, sizeof(Fq12));
        std::memcpy(&v.statement.final_A, s + 96, sizeof(G1Affine));
        std::memcpy(&v.statement.final_B, s + 112, sizeof(G2Affine));
        std::memcpy(&v.statement.final_Z, s + 144, sizeof(Fq12));
        return v;
    }

    sipp_ctx* ctx(int kind) { return ctx_[kind]; }

   private:
    sipp_ctx* ctx_[3] = {nullptr, nullptr, nullptr};
    size_t max_io_[3] = {0, 0, 0};
    bool verify_after_prove_ = true;

    template <class IO>
    static const uint32_t* words(const std::vector<IO>& v) {
        return reinterpret_cast<const uint32_t*>(v.data());
    }
    std::string first_error() const {
        for (sipp_ctx* c : ctx_) {
            const char* e = sipp_last_error(c);
            if (e && *e) return e;
        }
        return "";
    }
    template <class In, class Out>
    std::vector<ExpIO<In, Out>> outputs(int kind, const std::vector<In>& inputs) {
        if (inputs.empty() || inputs.size() > max_io_[kind]) throw Error(SIPP_E_BADARG, "exp_circuit: IO count out of range");
        std::vector<ExpIO<In, Out>> io(inputs.size());
        for (size_t i = 0; i < inputs.size(); i++) {
            io[i].in = inputs[i];
            std::memset(&io[i].out, 0, sizeof(Out));
        }
        const int rc = sipp_exp_outputs(ctx_[kind], kind, reinterpret_cast<uint32_t*>(io.data()), io.size());
        if (rc != SIPP_OK) throw Error(rc, std::string("sipp_exp_outputs: ") + sipp_last_error(ctx_[kind]));
        return io;
    }
    template <class In, class Out>
    void finish(const std::vector<ExpIO<In, Out>>& io, ExpCircuitResult<Out>* r) const {
        r->outputs.resize(io.size());
        for (size_t i = 0; i < io.size(); i++) r->outputs[i] = io[i].out;
        // the generators of starky-bn254 run starky's verify_stark_proof on the proof they have just made: the library's
        // own verifier, a few milliseconds on the host; set_verify_after_prove(false) leaves the check to the caller
        if (verify_after_prove_) verify_stark_proof(r->flat);
        r->proof = StarkProofWithPublicInputs::from_flat(r->flat.data(), r->flat.size());
    }
    template <class In, class Out>
    ExpCircuitResult<Out> run(int kind, const std::vector<In>& inputs, bool hardened = false) {
        std::vector<ExpIO<In, Out>> io = outputs<In, Out>(kind, inputs);
        ExpCircuitResult<Out> r;
        const int api_kind = hardened && kind < 2 ? kind + SIPP_G1_EXP_HARDENED : kind;
        const size_t cap = sipp_proof_size(ctx_[kind], api_kind, io.size());
        r.flat.assign(cap, 0);
        size_t len = 0;
        int rc;
        if (api_kind != kind)
            rc = sipp_prove(ctx_[kind], api_kind, words(io), io.size(), r.flat.data(), cap, &len);
        else if (kind == SIPP_G1_EXP)
            rc = sipp_g1_exp_prove(ctx_[kind], words(io), io.size(), r.flat.data(), cap, &len);
        else if (kind == SIPP_G2_EXP)
            rc = sipp_g2_exp_prove(ctx_[kind], words(io), io.size(), r.flat.data(), cap, &len);
        else
            rc = sipp_fq12_exp_prove(ctx_[kind], words(io), io.size(), r.flat.data(), cap, &len);
        if (rc != SIPP_OK) throw Error(rc, std::string("sipp_exp_prove: ") + sipp_last_error(ctx_[kind]));
        r.flat.resize(len);
        finish(io, &r);
        return r;
    }
};

}  // namespace sipp
