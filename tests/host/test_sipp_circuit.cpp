// tests/host/test_sipp_circuit.cpp -- the C++ host layer (include/sipp_host.hpp) driven the way the reference's own
// test drives the path (src/verifier_circuit.rs:192-269 `test_sipp_circuit`): build the three obligation lists, run
// g1_exp_circuit / g2_exp_circuit / fq12_exp_circuit, check the outputs against the native chain, verify the proofs.
//
//   test_sipp_circuit layout                      CPU only: record layouts, from_flat / to_flat round trip of <proof.bin>
//   test_sipp_circuit layout <proof.bin>
//   test_sipp_circuit fuzz <proof.bin> [n]        CPU only: from_flat on n damaged copies of the buffer (Error or round trip, never UB)
//   test_sipp_circuit prove <ios.bin> <out_prefix>   GPU: <ios.bin> = 3 x (u64 count, records) [+ (count, A), (count, B)];
//                                                    writes <out_prefix>{0,1,2}.bin
//   test_sipp_circuit mapg2 <msgs.bin> <out_prefix>  GPU: the BLS example's front end (src/bin/bls_aggregation.rs:65, :100-104):
//                                                    <msgs.bin> = (u64 count, count x 16 u32); writes <out_prefix>_proof.bin,
//                                                    <out_prefix>_points.bin (the cofactor-cleared points)
//
// Every proof goes through the library's own verifier (sipp::verify_stark_proof = data.verify's checks) AND through the CPU oracle's
// (test infrastructure, linked only into this test binary).
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <fstream>

#include "sipp_host.hpp"

struct orc_config {      // oracle/stark.h, field for field (the checker's configuration: orc_default_config fills all ten)
    uint32_t rate_bits, cap_height, pow_bits, arity_bits, final_poly_bits, num_queries, num_challenges, pow_rule, fs_rule, lookup_rule;
};
extern "C" void orc_default_config(orc_config* c);
extern "C" int orc_stark_verify(const uint64_t* proof, size_t len, const orc_config* cfg);

static std::vector<uint64_t> read_u64(const char* path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) {
        fprintf(stderr, "cannot open %s\n", path);
        exit(2);
    }
    const size_t bytes = (size_t)f.tellg();
    std::vector<uint64_t> v(bytes / 8);
    f.seekg(0);
    f.read(reinterpret_cast<char*>(v.data()), (std::streamsize)(v.size() * 8));
    return v;
}

#define CHECK(c)                                                        \
    do {                                                                \
        if (!(c)) {                                                     \
            fprintf(stderr, "%s:%d: CHECK failed: %s\n", __FILE__, __LINE__, #c); \
            return 1;                                                   \
        }                                                               \
    } while (0)

static int layout(const char* proof_path) {
    CHECK(sizeof(sipp::G1ExpInput) == 4 * (SIPP_G1_IO_WORDS - 16));
    CHECK(sizeof(sipp::G2ExpInput) == 4 * (SIPP_G2_IO_WORDS - 32));
    CHECK(sizeof(sipp::Fq12ExpInput) == 4 * (SIPP_FQ12_IO_WORDS - 96));
    CHECK(offsetof(sipp::G1ExpIO, out) == 4 * 40 && offsetof(sipp::G2ExpIO, out) == 4 * 72 && offsetof(sipp::Fq12ExpIO, out) == 4 * 200);
    {
        // level L-D: the per-GPU slices of an obligation list tile it (host arithmetic of the C ABI, no GPU)
        std::vector<sipp::G1ExpInput> all(127);
        for (size_t i = 0; i < all.size(); i++) memset(&all[i], (int)i, sizeof(all[i]));
        for (unsigned world : {1u, 2u, 8u, 200u}) {
            size_t seen = 0;
            for (unsigned rank = 0; rank < world; rank++) {
                const auto mine = sipp::Prover::io_shard(all, world, rank);
                CHECK(mine.size() <= (all.size() + world - 1) / world);
                for (size_t j = 0; j < mine.size(); j++) CHECK(memcmp(&mine[j], &all[seen + j], sizeof(all[0])) == 0);
                seen += mine.size();
            }
            CHECK(seen == all.size());
        }
        try {
            (void)sipp::Prover::io_shard(all, 4, 4);
            CHECK(!"rank == world accepted");
        } catch (const sipp::Error& e) {
            CHECK(e.status() == SIPP_E_BADARG);
        }
    }
    if (!proof_path) return 0;
    const std::vector<uint64_t> flat = read_u64(proof_path);
    const auto p = sipp::StarkProofWithPublicInputs::from_flat(flat.data(), flat.size());
    CHECK(p.proof.trace_cap.size() == (size_t)1 << p.cap_height);
    CHECK(p.proof.openings.local_values.size() == p.main_cols && p.proof.openings.permutation_zs_next.size() == p.perm_cols);
    CHECK(p.proof.opening_proof.query_round_proofs.size() == 84);
    CHECK(p.proof.opening_proof.query_round_proofs[0].initial_trees_proof.evals_proofs.size() == 3);
    CHECK(p.public_inputs.size() == (size_t)p.num_io * p.pi_per_io);
    CHECK(p.to_flat() == flat);
    // a truncated or foreign buffer is refused with an Error, not undefined behaviour
    try {
        (void)sipp::StarkProofWithPublicInputs::from_flat(flat.data(), flat.size() - 1);
        CHECK(!"truncated buffer accepted");
    } catch (const sipp::Error& e) {
        CHECK(e.status() == SIPP_E_BADARG || e.status() == SIPP_E_BUFSZ);
    }
    printf("layout ok: %zu words, %u main / %u Z / %u quotient columns\n", flat.size(), p.main_cols, p.perm_cols, p.quotient_cols);
    return 0;
}

template <class IO, class In>
static std::vector<In> inputs_of(const std::vector<IO>& io) {
    std::vector<In> v(io.size());
    for (size_t i = 0; i < io.size(); i++) v[i] = io[i].in;
    return v;
}

template <class IO>
static std::vector<IO> take(const std::vector<uint64_t>& w, size_t* pos) {
    const size_t cnt = (size_t)w[(*pos)++];
    std::vector<IO> v(cnt);
    std::memcpy(v.data(), w.data() + *pos, cnt * sizeof(IO));
    *pos += (cnt * sizeof(IO) + 7) / 8;
    return v;
}

// from_flat on damaged buffers: a proof buffer may come from anywhere, so the parser must answer every mutation with an
// Error (or a successful parse that round-trips), never with undefined behaviour -- run under AddressSanitizer + UBSan by
// scripts/run_asan.sh.  Mutations: header fields set to extreme values, random words flipped, truncations, extensions.
static int fuzz(const char* proof_path, int iterations) {
    const std::vector<uint64_t> flat = read_u64(proof_path);
    uint64_t x = 0x9E3779B97F4A7C15ULL;
    auto rnd = [&]() {
        x ^= x << 13;
        x ^= x >> 7;
        x ^= x << 17;
        return x;
    };
    const uint64_t extreme[] = {0, 1, 2, 31, 32, 33, 63, 64, 255, 65535, 0x7fffffffULL, 0xffffffffULL, 0x100000000ULL, ~0ULL, ~0ULL >> 1};
    size_t parsed = 0, refused = 0;
    for (int it = 0; it < iterations; it++) {
        std::vector<uint64_t> b = flat;
        const int mode = (int)(rnd() % 5);
        if (mode == 0) {
            b[1 + rnd() % 15] = extreme[rnd() % (sizeof extreme / sizeof extreme[0])];
        } else if (mode == 1) {
            for (int k = 0; k < 1 + (int)(rnd() % 4); k++) b[1 + rnd() % 15] = extreme[rnd() % (sizeof extreme / sizeof extreme[0])];
            b[12] = b.size();
        } else if (mode == 2) {
            b[rnd() % b.size()] ^= 1ULL << (rnd() % 64);
        } else if (mode == 3) {
            b.resize(rnd() % b.size());
            if (b.size() > 12 && (rnd() & 1)) b[12] = b.size();
        } else {
            b.resize(b.size() + 1 + rnd() % 64, rnd());
            if (rnd() & 1) b[12] = b.size();
        }
        try {
            const auto p = sipp::StarkProofWithPublicInputs::from_flat(b.data(), b.size());
            CHECK(p.to_flat() == b);     // whatever parses must re-serialise to the same words
            parsed++;
        } catch (const sipp::Error&) {
            refused++;
        }
    }
    printf("fuzz ok: %d mutations, %zu refused, %zu parsed and round-tripped\n", iterations, refused, parsed);
    return 0;
}

static int prove(const char* ios_path, const char* out_prefix) {
    const std::vector<uint64_t> w = read_u64(ios_path);
    size_t pos = 0;
    const auto io1 = take<sipp::G1ExpIO>(w, &pos);
    const auto io2 = take<sipp::G2ExpIO>(w, &pos);
    const auto io12 = take<sipp::Fq12ExpIO>(w, &pos);
    sipp::Prover prover(0, io1.size(), io2.size(), io12.size());

    // optional fourth section: the points A, B themselves -> the native chain in front of the circuit
    // (verifier_circuit.rs:202-211: sipp_prove_native, then sipp_verify_native gives the statement and the obligations)
    if (pos < w.size()) {
        const auto A = take<sipp::G1Affine>(w, &pos);
        const auto B = take<sipp::G2Affine>(w, &pos);
        const std::vector<sipp::Fq12> sipp_proof = prover.sipp_prove_native(A, B);
        const auto v = prover.sipp_verify_native(A, B, sipp_proof);
        CHECK(v.g1_obligations.size() == io1.size() && v.g2_obligations.size() == io2.size() && v.fq12_obligations.size() == io12.size());
        CHECK(std::memcmp(v.g1_obligations.data(), io1.data(), io1.size() * sizeof(sipp::G1ExpIO)) == 0);
        CHECK(std::memcmp(v.g2_obligations.data(), io2.data(), io2.size() * sizeof(sipp::G2ExpIO)) == 0);
        CHECK(std::memcmp(v.fq12_obligations.data(), io12.data(), io12.size() * sizeof(sipp::Fq12ExpIO)) == 0);
        CHECK(std::memcmp(&v.statement.final_Z, &io12.back().out, sizeof(sipp::Fq12)) == 0);
        auto bad = sipp_proof;
        bad[0].c[0][0] ^= 1;  // final message Z_R of the last round
        try {
            (void)prover.sipp_verify_native(A, B, bad);
            CHECK(!"tampered SIPP proof accepted");
        } catch (const sipp::Error& e) {
            CHECK(e.status() == SIPP_E_WITNESS);
        }
        printf("native chain ok: %zu pairs, %zu proof messages\n", A.size(), sipp_proof.size());
        // the BLS example's last step (src/bin/bls_aggregation.rs:76-77): z = pairing_circuit(final_A, final_B), connected to final_Z
        const auto pr = prover.pairing_circuit(v.statement.final_A, v.statement.final_B);
        CHECK(pr.outputs.size() == 1 && std::memcmp(&pr.outputs[0], &v.statement.final_Z, sizeof(sipp::Fq12)) == 0);
        orc_config pcfg;
        orc_default_config(&pcfg);
        sipp::verify_stark_proof(pr.flat);                                        // the library's verifier ...
        CHECK(orc_stark_verify(pr.flat.data(), pr.flat.size(), &pcfg) == 0);       // ... and the oracle's
        CHECK(pr.proof.kind == (uint32_t)SIPP_PAIRING && pr.proof.to_flat() == pr.flat);
        // the public inputs are the record (final_A, final_B, final_Z), padded by a copy
        const uint32_t* fa = reinterpret_cast<const uint32_t*>(&v.statement.final_A);
        for (size_t i = 0; i < 16; i++) CHECK(pr.proof.public_inputs[i] == fa[i]);
        const uint32_t* fz = reinterpret_cast<const uint32_t*>(&v.statement.final_Z);
        for (size_t i = 0; i < 96; i++) CHECK(pr.proof.public_inputs[48 + i] == fz[i]);
        printf("final pairing ok: proof of %zu words verified\n", pr.flat.size());
    }

    // the three calls of verifier_circuit.rs:133-135
    sipp::ExpCircuitResult<sipp::G1Affine> r1;
    sipp::ExpCircuitResult<sipp::G2Affine> r2;
    sipp::ExpCircuitResult<sipp::Fq12> r12;
    prover.exp_circuits(inputs_of<sipp::G1ExpIO, sipp::G1ExpInput>(io1), inputs_of<sipp::G2ExpIO, sipp::G2ExpInput>(io2),
                        inputs_of<sipp::Fq12ExpIO, sipp::Fq12ExpInput>(io12), &r1, &r2, &r12);
    // outputs == the native chain's values (the fixture's output words, oracle/py/sipp_native.py)
    for (size_t i = 0; i < io1.size(); i++) CHECK(std::memcmp(&r1.outputs[i], &io1[i].out, sizeof(sipp::G1Affine)) == 0);
    for (size_t i = 0; i < io2.size(); i++) CHECK(std::memcmp(&r2.outputs[i], &io2[i].out, sizeof(sipp::G2Affine)) == 0);
    for (size_t i = 0; i < io12.size(); i++) CHECK(std::memcmp(&r12.outputs[i], &io12[i].out, sizeof(sipp::Fq12)) == 0);
    // data.verify(proof) of verifier_circuit.rs:254, per sub-proof
    orc_config ocfg;
    orc_default_config(&ocfg);
    const std::vector<uint64_t>* flats[3] = {&r1.flat, &r2.flat, &r12.flat};
    const sipp::StarkProofWithPublicInputs* proofs[3] = {&r1.proof, &r2.proof, &r12.proof};
    for (int k = 0; k < 3; k++) {
        sipp::verify_stark_proof(*flats[k]);
        CHECK(orc_stark_verify(flats[k]->data(), flats[k]->size(), &ocfg) == 0);
        {   // a damaged proof is refused by both, at the same stage
            std::vector<uint64_t> bad = *flats[k];
            bad[bad.size() / 2] ^= 1;
            int stage = 0;
            CHECK(sipp_stark_verify(bad.data(), bad.size(), nullptr, &stage) == SIPP_E_VERIFY);
            CHECK(stage == -orc_stark_verify(bad.data(), bad.size(), &ocfg));
        }
        CHECK(proofs[k]->to_flat() == *flats[k]);
        CHECK(proofs[k]->kind == (uint32_t)k);
        std::ofstream f(std::string(out_prefix) + char('0' + k) + ".bin", std::ios::binary);
        f.write(reinterpret_cast<const char*>(flats[k]->data()), (std::streamsize)(flats[k]->size() * 8));
    }
    // the single-call form gives the same proof, and public inputs are exactly the records
    const auto again = prover.g1_exp_circuit(inputs_of<sipp::G1ExpIO, sipp::G1ExpInput>(io1));
    CHECK(again.flat == r1.flat);
    const uint32_t* rec = reinterpret_cast<const uint32_t*>(io1.data());
    for (size_t i = 0; i < io1.size() * SIPP_G1_IO_WORDS; i++) CHECK(r1.proof.public_inputs[i] == rec[i]);
    // error behaviour: an unprovable obligation (x = offset, odd exponent) throws with SIPP_E_WITNESS; the prover survives
    {
        auto bad = inputs_of<sipp::G1ExpIO, sipp::G1ExpInput>(io1);
        bad[0].offset = bad[0].x;
        bad[0].exp_val[0] |= 1;
        try {
            (void)prover.g1_exp_circuit(bad);
            CHECK(!"degenerate obligation accepted");
        } catch (const sipp::Error& e) {
            CHECK(e.status() == SIPP_E_WITNESS);
        }
        CHECK(prover.g1_exp_circuit(inputs_of<sipp::G1ExpIO, sipp::G1ExpInput>(io1)).flat == r1.flat);
    }
    printf("test_sipp_circuit ok: %zu / %zu / %zu obligations, proofs of %zu / %zu / %zu words verified\n", io1.size(), io2.size(),
           io12.size(), r1.flat.size(), r2.flat.size(), r12.flat.size());
    return 0;
}

// messages -> batch_map_to_g2_circuit (proof verified by the oracle, public inputs = (u, x, y) in message order, padded by copies of
// the last record) -> map_to_g2 (cleared points) -> the cofactor obligations through g2_exp_circuit: outputs = the cleared points
static int mapg2(const char* msgs_path, const char* out_prefix) {
    const std::vector<uint64_t> raw = read_u64(msgs_path);
    CHECK(!raw.empty());
    const size_t n = (size_t)raw[0];
    CHECK(raw.size() >= 1 + n * 8);
    std::vector<sipp::Fq2> msgs(n);
    memcpy(msgs.data(), raw.data() + 1, n * sizeof(sipp::Fq2));
    sipp::Prover prover(0, 2, 2 * n, 2);
    const auto r = prover.batch_map_to_g2_circuit(msgs);
    CHECK(r.outputs.size() == n);
    orc_config cfg;
    orc_default_config(&cfg);
    sipp::verify_stark_proof(r.flat);
    CHECK(orc_stark_verify(r.flat.data(), r.flat.size(), &cfg) == 0);
    CHECK(r.proof.public_inputs.size() % SIPP_MAP_G2_IO_WORDS == 0 && r.proof.public_inputs.size() / SIPP_MAP_G2_IO_WORDS >= n);
    for (size_t i = 0; i < n; i++) {
        sipp::MapG2IO rec;
        rec.in = msgs[i];
        rec.out = r.outputs[i];
        const uint32_t* w = reinterpret_cast<const uint32_t*>(&rec);
        for (int k = 0; k < SIPP_MAP_G2_IO_WORDS; k++) CHECK(r.proof.public_inputs[i * SIPP_MAP_G2_IO_WORDS + k] == w[k]);
    }
    const auto m = prover.map_to_g2(msgs);
    CHECK(m.points.size() == n && m.cofactor_inputs.size() == 2 * n);
    for (size_t i = 0; i < n; i++) CHECK(memcmp(&m.cofactor_inputs[i].x, &r.outputs[i], sizeof(sipp::G2Affine)) == 0);
    const auto c = prover.g2_exp_circuit(m.cofactor_inputs);
    sipp::verify_stark_proof(c.flat);
    CHECK(orc_stark_verify(c.flat.data(), c.flat.size(), &cfg) == 0);
    {   // the same obligations through the hardened AIR: another proof (kind 5 in its header), the same outputs
        const auto ch = prover.g2_exp_circuit(m.cofactor_inputs, true);
        CHECK(ch.flat.size() > c.flat.size() && ch.flat[1] == SIPP_G2_EXP_HARDENED);
        CHECK(orc_stark_verify(ch.flat.data(), ch.flat.size(), &cfg) == 0);
        for (size_t i = 0; i < 2 * n; i++) CHECK(memcmp(&ch.outputs[i], &c.outputs[i], sizeof(sipp::G2Affine)) == 0);
    }
    for (size_t i = 0; i < n; i++) CHECK(memcmp(&c.outputs[n + i], &m.points[i], sizeof(sipp::G2Affine)) == 0);
    std::ofstream(std::string(out_prefix) + "_proof.bin", std::ios::binary).write(reinterpret_cast<const char*>(r.flat.data()), (std::streamsize)(r.flat.size() * 8));
    std::ofstream(std::string(out_prefix) + "_points.bin", std::ios::binary).write(reinterpret_cast<const char*>(m.points.data()), (std::streamsize)(n * sizeof(sipp::G2Affine)));
    printf("mapg2 ok: %zu messages, MapToG2 proof %zu words, cofactor proof %zu words\n", n, r.flat.size(), c.flat.size());
    return 0;
}

// The tail of the reference's test (src/verifier_circuit.rs:225, :253-268) in its own terms: build the circuit data once, `data.prove(pw)`,
// `data.verify(proof)`, read the public inputs back.  The circuit is DATA (fixture written by tests/test_host_cpp.py from
// tools/plonk_synth.py: the recursion-shaped stand-in, chained): u64 words
//   header[12] = degree_bits, num_wires, num_routed, num_constants, num_selectors, num_gates, program_words, n_gens, n_levels, n_copies,
//                n_public_inputs, max_degree | fri params (7 + 32 words) | gates (6 words each) | programs | generators (8 words each)
//   | rows (N) | level_offsets | copy_src | copy_dst | copy_offsets | constants_sigmas | wires (input cells set) | public inputs
static int outer(const char* path, const char* out_path) {
    const std::vector<uint64_t> w = read_u64(path);
    size_t pos = 0;
    auto next = [&](size_t count) -> const uint64_t* {
        if (pos + count > w.size()) {
            fprintf(stderr, "circuit fixture too short\n");
            exit(2);
        }
        const uint64_t* q = w.data() + pos;
        pos += count;
        return q;
    };
    const uint64_t* h = next(12);
    const uint32_t degree_bits = (uint32_t)h[0], n_gens = (uint32_t)h[7], n_levels = (uint32_t)h[8], n_copies = (uint32_t)h[9], n_pi = (uint32_t)h[10];
    const size_t n = (size_t)1 << degree_bits;
    sipp_plonk_params params{(uint32_t)h[2], (uint32_t)h[11], 2};
    sipp_fri_params fri{};
    const uint64_t* f = next(7 + SIPP_FRI_MAX_ROUNDS);
    fri.rate_bits = (uint32_t)f[0]; fri.cap_height = (uint32_t)f[1]; fri.pow_bits = (uint32_t)f[2]; fri.num_queries = (uint32_t)f[3];
    fri.pow_rule = (uint32_t)f[4]; fri.hiding = (uint32_t)f[5]; fri.n_rounds = (uint32_t)f[6];
    for (uint32_t i = 0; i < SIPP_FRI_MAX_ROUNDS; i++) fri.arity_bits[i] = (uint32_t)f[7 + i];
    std::vector<sipp_plonk_gate> gates(h[5]);
    for (auto& g : gates) {
        const uint64_t* q = next(6);
        g = sipp_plonk_gate{(uint32_t)q[0], (uint32_t)q[1], (uint32_t)q[2], (uint32_t)q[3], (uint32_t)q[4], (uint32_t)q[5]};
    }
    const uint64_t* pr = next(h[6]);
    std::vector<int64_t> programs(pr, pr + h[6]);
    std::vector<sipp_plonk_generator> gens(n_gens);
    for (auto& g : gens) {
        const uint64_t* q = next(8);
        g = sipp_plonk_generator{(uint32_t)q[0], (uint32_t)q[1], (uint32_t)q[2], {(uint32_t)q[3], (uint32_t)q[4], (uint32_t)q[5], (uint32_t)q[6], (uint32_t)q[7]}};
    }
    auto u32s = [&](size_t count) {
        const uint64_t* q = next(count);
        return std::vector<uint32_t>(q, q + count);
    };
    const std::vector<uint32_t> rows = u32s(n), level_offsets = u32s(n_levels + 1);
    const uint64_t* cs_ = next(n_copies);
    const uint64_t* cd_ = next(n_copies);
    const std::vector<uint32_t> copy_offsets = u32s(n_levels + 1);
    const sipp_plonk_schedule_host sched{n_levels, rows.data(), level_offsets.data(), cs_, cd_, copy_offsets.data()};
    const size_t cs_words = (size_t)(h[3] + h[2]) * n, wire_words = (size_t)h[1] * n;
    const uint64_t* q = next(cs_words);
    const std::vector<uint64_t> constants_sigmas(q, q + cs_words);
    q = next(wire_words);
    const std::vector<uint64_t> pw(q, q + wire_words);
    q = next(n_pi);
    const std::vector<uint64_t> statement(q, q + n_pi);
    CHECK(pos == w.size());
    const sipp_plonk_circuit circuit{(uint32_t)h[1], (uint32_t)h[3], (uint32_t)h[4], (uint32_t)gates.size(), gates.data(), programs.data(),
                                     (uint32_t)programs.size()};

    sipp::CircuitData data(0, degree_bits, params, fri, circuit, constants_sigmas, gens, &sched);     // builder.build::<C>()
    const sipp::ProofWithPublicInputs proof = data.prove(pw, statement);                              // data.prove(pw)
    data.verify(proof);                                                                               // data.verify(proof)
    CHECK(proof.public_inputs == statement);                      // SIPPStatement::from_vec(&proof.public_inputs) == statement (:258-268)
    sipp::verify_plonk_proof(proof.flat, data.constants_sigmas_cap, params, fri, circuit, data.circuit_digest);   // ... from the verifier data alone
    sipp::ProofWithPublicInputs forged = proof;
    forged.flat.back() ^= 1;
    try {
        data.verify(forged);
        CHECK(!"a proof with another statement was accepted");
    } catch (const sipp::Error& e) {
        CHECK(e.status() == SIPP_E_VERIFY);
    }
    std::ofstream o(out_path, std::ios::binary);
    o.write(reinterpret_cast<const char*>(proof.flat.data()), (std::streamsize)(proof.flat.size() * 8));
    o.write(reinterpret_cast<const char*>(data.circuit_digest), 32);
    printf("outer proof ok: %zu words, %u public inputs, %u levels\n", proof.flat.size(), n_pi, n_levels);
    return 0;
}

int main(int argc, char** argv) {
    try {
        if (argc == 4 && std::string(argv[1]) == "outer") return outer(argv[2], argv[3]);
        if (argc >= 2 && std::string(argv[1]) == "layout") return layout(argc >= 3 ? argv[2] : nullptr);
        if (argc == 4 && std::string(argv[1]) == "prove") return prove(argv[2], argv[3]);
        if (argc == 4 && std::string(argv[1]) == "mapg2") return mapg2(argv[2], argv[3]);
        if (argc >= 3 && std::string(argv[1]) == "fuzz") return fuzz(argv[2], argc >= 4 ? atoi(argv[3]) : 20000);
    } catch (const sipp::Error& e) {
        fprintf(stderr, "sipp::Error %d: %s\n", e.status(), e.what());
        return 1;
    }
    fprintf(stderr, "usage: %s layout [proof.bin] | fuzz <proof.bin> [iterations] | prove <ios.bin> <out_prefix>\n", argv[0]);
    return 2;
}
