// tests/host/verify_fuzz.cpp -- the library's verifier (sipp_amd/csrc/verify.cpp) on DAMAGED input under AddressSanitizer / UBSan.
// A verifier reads untrusted words: whatever the buffer holds, it must end in a verdict -- no out-of-bounds read, no undefined shift, no
// allocation driven by an unchecked header field.  verify.cpp and the host permutation are compiled INTO this binary with the sanitizers
// (the shipped library is built without them); the two symbols verify.cpp takes from HIP translation units are restated here.
//   verify_fuzz <proof.bin> [iterations = 20000] [seed]        a STARK proof (sipp_stark_verify)
//   verify_fuzz case <case.bin> [iterations = 5000] [seed]      an opening proof / an outer proof with its verifier data (see run_case)
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <random>
#include <string>
#include <vector>

#include "sipp_hip.h"
#include "air_tables.h"

// trace.hip: the AIR of an API kind at a trace length (hardened kinds 4 / 5 = the hardened tables of kinds 0 / 1; u16 cells from 2^16 rows)
const air_spec_t* sipp_air_get(int kind, uint32_t log_n) {
    if (kind < 0 || kind > 6) return nullptr;
    const int hard = kind == 4 || kind == 5, base = hard ? kind - 4 : kind;
    const bool u16 = log_n >= 16;
    for (size_t i = 0; i < sizeof(AIR_AIRS) / sizeof(AIR_AIRS[0]); i++)
        if (AIR_AIRS[i].kind == base && AIR_AIRS[i].hardened == hard && (AIR_AIRS[i].table_bits == 16) == u16) return &AIR_AIRS[i];
    return nullptr;
}
// api.hip
extern "C" void sipp_default_config(sipp_stark_config* c) {
    c->rate_bits = 1; c->cap_height = 4; c->pow_bits = 16; c->arity_bits = 4; c->final_poly_bits = 5;
    c->num_queries = 84; c->num_challenges = 2; c->pow_rule = 0; c->fs_rule = 0; c->lookup_rule = 0;
}

// ---- the generic verifiers on a serialized case (tests/test_product_verifier_generic.py writes them: u64 words) -----------------------
//   kind 1 (sipp_fri_verify_openings): 1, log_n, n_oracles, n_batches, [rate_bits, cap_height, pow_bits, num_queries, pow_rule, hiding, n_rounds,
//       arity x 32], per oracle (ncols, n_salt, cap words), per batch (c0, c1, n_ranges, (oracle, begin, end) x n_ranges), challenger
//       (state 12, in 8, n_in, out 8, n_out), proof_len, proof
//   kind 2 (sipp_plonk_verify_gates): 2, R, D, C, [fri params as above], num_wires, num_constants, num_selectors, num_gates, gates x 6,
//       program_words, programs, digest 4, constants_sigmas cap, proof_len, proof
static std::vector<uint64_t> read_words(const char* path) {
    std::vector<uint64_t> w;
    FILE* f = fopen(path, "rb");
    if (!f) return w;
    fseek(f, 0, SEEK_END);
    const long bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    w.resize((size_t)bytes / 8);
    if (fread(w.data(), 8, w.size(), f) != w.size()) w.clear();
    fclose(f);
    return w;
}
static int run_case(const char* path, long iters, uint64_t seed) {
    const std::vector<uint64_t> w = read_words(path);
    if (w.size() < 64) return 2;
    size_t at = 0;
    auto next = [&]() { return w[at++]; };
    auto fri = [&](sipp_fri_params& fp) {
        fp.rate_bits = (uint32_t)next(); fp.cap_height = (uint32_t)next(); fp.pow_bits = (uint32_t)next(); fp.num_queries = (uint32_t)next();
        fp.pow_rule = (uint32_t)next(); fp.hiding = (uint32_t)next(); fp.n_rounds = (uint32_t)next();
        for (int i = 0; i < 32; i++) fp.arity_bits[i] = (uint32_t)next();
    };
    const uint64_t kind = next();
    std::mt19937_64 rng(seed);
    long refused = 0, accepted = 0;
    auto damage = [&](const std::vector<uint64_t>& proof) {
        std::vector<uint64_t> bad = proof;
        const int how = (int)(rng() % 6);
        if (how == 0) bad[rng() % 16] = rng() % 40;                                  // a header word
        else if (how == 1) bad.resize(8 + rng() % (bad.size() - 8));                  // truncation
        else if (how == 2) { bad.resize(8 + rng() % (bad.size() - 8)); bad[kind == 1 ? 6 : 5] = bad.size(); }   // ... with a consistent length word
        else if (how == 3) bad[rng() % bad.size()] = rng();
        else if (how == 4) bad[rng() % bad.size()] = rng() % 0xFFFFFFFF00000001ull;
        else bad.resize(bad.size() + 1 + rng() % 9, 1);
        return bad;
    };
    if (kind == 1) {
        const uint32_t log_n = (uint32_t)next();
        const size_t n_or = (size_t)next(), n_b = (size_t)next();
        sipp_fri_params fp;
        fri(fp);
        std::vector<uint32_t> ncols(n_or), n_salt(n_or);
        std::vector<std::vector<uint64_t>> caps(n_or);
        std::vector<const uint64_t*> cptr(n_or);
        for (size_t o = 0; o < n_or; o++) {
            ncols[o] = (uint32_t)next(); n_salt[o] = (uint32_t)next();
            for (size_t i = 0; i < ((size_t)4 << fp.cap_height); i++) caps[o].push_back(next());
            cptr[o] = caps[o].data();
        }
        std::vector<sipp_fri_batch> batches(n_b);
        std::vector<std::vector<sipp_poly_range>> ranges(n_b);
        for (size_t b = 0; b < n_b; b++) {
            batches[b].point[0] = next(); batches[b].point[1] = next();
            batches[b].n_ranges = (uint32_t)next();
            for (uint32_t r = 0; r < batches[b].n_ranges; r++) {
                sipp_poly_range rg;
                rg.oracle = (uint32_t)next(); rg.col_begin = (uint32_t)next(); rg.col_end = (uint32_t)next();
                ranges[b].push_back(rg);
            }
            batches[b].ranges = ranges[b].data();
        }
        sipp_challenger ch0;
        for (int i = 0; i < 12; i++) ch0.state[i] = next();
        for (int i = 0; i < 8; i++) ch0.in_buf[i] = next();
        ch0.n_in = next();
        for (int i = 0; i < 8; i++) ch0.out_buf[i] = next();
        ch0.n_out = next();
        const size_t plen = (size_t)next();
        const std::vector<uint64_t> proof(w.begin() + (long)at, w.begin() + (long)(at + plen));
        int stage = -1;
        sipp_challenger ch = ch0;
        if (sipp_fri_verify_openings(proof.data(), proof.size(), cptr.data(), ncols.data(), n_salt.data(), n_or, batches.data(), n_b, log_n, &fp, &ch,
                                     &stage) != SIPP_OK) {
            fprintf(stderr, "the undamaged opening proof is refused at stage %d\n", stage);
            return 1;
        }
        for (long it = 0; it < iters; it++) {
            const std::vector<uint64_t> bad = damage(proof);
            if (bad == proof) continue;
            ch = ch0;
            const int rc = sipp_fri_verify_openings(bad.data(), bad.size(), cptr.data(), ncols.data(), n_salt.data(), n_or, batches.data(), n_b, log_n,
                                                    &fp, &ch, &stage);
            if (rc == SIPP_OK) accepted++;
            else if (rc == SIPP_E_VERIFY && stage >= 100 && stage <= 141) refused++;
            else { fprintf(stderr, "unexpected verdict %d / stage %d\n", rc, stage); return 1; }
        }
    } else if (kind == 2) {
        sipp_plonk_params pp;
        pp.num_routed_wires = (uint32_t)next(); pp.max_degree = (uint32_t)next(); pp.num_challenges = (uint32_t)next();
        sipp_fri_params fp;
        fri(fp);
        sipp_plonk_circuit c;
        c.num_wires = (uint32_t)next(); c.num_constants = (uint32_t)next(); c.num_selectors = (uint32_t)next(); c.num_gates = (uint32_t)next();
        std::vector<sipp_plonk_gate> gates(c.num_gates);
        for (auto& g : gates) {
            g.selector_index = (uint32_t)next(); g.row = (uint32_t)next(); g.group_lo = (uint32_t)next(); g.group_hi = (uint32_t)next();
            g.prog_offset = (uint32_t)next(); g.num_constraints = (uint32_t)next();
        }
        c.gates = gates.data();
        c.program_words = (uint32_t)next();
        std::vector<int64_t> prog(c.program_words);
        for (auto& x : prog) x = (int64_t)next();
        c.programs = prog.data();
        uint64_t digest[4];
        for (auto& x : digest) x = next();
        std::vector<uint64_t> cap((size_t)4 << fp.cap_height);
        for (auto& x : cap) x = next();
        const size_t plen = (size_t)next();
        const std::vector<uint64_t> proof(w.begin() + (long)at, w.begin() + (long)(at + plen));
        int stage = -1;
        if (sipp_plonk_verify_gates(proof.data(), proof.size(), cap.data(), &pp, &fp, &c, digest, &stage) != SIPP_OK) {
            fprintf(stderr, "the undamaged outer proof is refused at stage %d\n", stage);
            return 1;
        }
        for (long it = 0; it < iters; it++) {
            const std::vector<uint64_t> bad = damage(proof);
            if (bad == proof) continue;
            const int rc = sipp_plonk_verify_gates(bad.data(), bad.size(), cap.data(), &pp, &fp, &c, digest, &stage);
            if (rc == SIPP_OK) accepted++;
            else if (rc == SIPP_E_VERIFY && stage >= 100 && stage <= 210) refused++;
            else { fprintf(stderr, "unexpected verdict %d / stage %d\n", rc, stage); return 1; }
        }
    } else {
        return 2;
    }
    printf("verify fuzz ok (case kind %llu): %ld damaged proofs refused, %ld accepted\n", (unsigned long long)kind, refused, accepted);
    return accepted == 0 ? 0 : 1;
}

int main(int argc, char** argv) {
    if (argc >= 3 && std::string(argv[1]) == "case")
        return run_case(argv[2], argc > 3 ? atol(argv[3]) : 5000, argc > 4 ? (uint64_t)atoll(argv[4]) : 99);
    if (argc < 2) {
        fprintf(stderr, "usage: %s <proof.bin> [iterations] [seed]\n", argv[0]);
        return 2;
    }
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    fseek(f, 0, SEEK_END);
    const long bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint64_t> proof((size_t)bytes / 8);
    if (fread(proof.data(), 8, proof.size(), f) != proof.size()) return 2;
    fclose(f);
    const long iters = argc > 2 ? atol(argv[2]) : 20000;
    std::mt19937_64 rng(argc > 3 ? (uint64_t)atoll(argv[3]) : 12345);
    int stage = -1;
    if (sipp_stark_verify(proof.data(), proof.size(), nullptr, &stage) != SIPP_OK) {
        fprintf(stderr, "the undamaged proof is refused at stage %d\n", stage);
        return 1;
    }
    long refused = 0, accepted = 0;
    long by_stage[200] = {0};
    const uint64_t extremes[] = {0, 1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 12, 13, 14, 16, 26, 27, 31, 32, 63, 64, 255, 1u << 16, 0xffffffffull, 1ull << 32, 1ull << 40, ~0ull,
                                 0xFFFFFFFF00000000ull, 0xFFFFFFFF00000001ull};
    for (long it = 0; it < iters; it++) {
        std::vector<uint64_t> bad = proof;
        const int how = (int)(rng() % 8);
        if (how == 0) {                         // a header field at an extreme value (shape, sizes, counts)
            bad[1 + rng() % 15] = extremes[rng() % (sizeof extremes / 8)];
        } else if (how == 1) {                  // ... and the length word made consistent with a truncation
            const size_t cut = 1 + rng() % (bad.size() - 17);
            bad.resize(bad.size() - cut);
            bad[12] = bad.size();
        } else if (how == 2) {                  // plain truncation / extension
            if (rng() & 1) bad.resize(16 + rng() % (bad.size() - 16));
            else bad.resize(bad.size() + 1 + rng() % 64, rng());
        } else if (how == 3) {                  // a few random words anywhere
            for (int k = 0; k < 1 + (int)(rng() % 4); k++) bad[rng() % bad.size()] = rng();
        } else if (how == 4) {                  // a random canonical field element
            bad[16 + rng() % (bad.size() - 16)] = rng() % 0xFFFFFFFF00000001ull;
        } else if (how == 5) {                  // a public-input word (record elements: canonical form, curve membership)
            const size_t npi = (size_t)(proof[3] * proof[11]);
            bad[bad.size() - 1 - rng() % npi] = rng() & 0xffffffffu;
        } else if (how == 6) {                  // two header fields at once
            bad[1 + rng() % 15] = extremes[rng() % (sizeof extremes / 8)];
            bad[1 + rng() % 15] = extremes[rng() % (sizeof extremes / 8)];
        } else {                                // a block of words shifted by one
            const size_t at = 16 + rng() % (bad.size() - 32), cnt = 1 + rng() % 15;
            memmove(&bad[at], &bad[at + 1], cnt * 8);
        }
        if (bad == proof) continue;
        stage = -1;
        const int rc = sipp_stark_verify(bad.data(), bad.size(), nullptr, &stage);
        if (rc == SIPP_OK) accepted++;
        else {
            refused++;
            if (rc != SIPP_E_VERIFY || stage < 100 || stage > 141) {
                fprintf(stderr, "unexpected verdict %d / stage %d at iteration %ld\n", rc, stage, it);
                return 1;
            }
            by_stage[stage]++;
        }
    }
    printf("verify fuzz ok: %ld damaged proofs refused, %ld accepted; stages:", refused, accepted);
    for (int s = 100; s < 142; s++)
        if (by_stage[s]) printf(" %d:%ld", s, by_stage[s]);
    printf("\n");
    return accepted == 0 ? 0 : 1;
}
