// tests/host/verify_fuzz.cpp -- the library's verifier (sipp_amd/csrc/verify.cpp) on DAMAGED input under AddressSanitizer / UBSan.
// A verifier reads untrusted words: whatever the buffer holds, it must end in a verdict -- no out-of-bounds read, no undefined shift, no
// allocation driven by an unchecked header field.  verify.cpp and the host permutation are compiled INTO this binary with the sanitizers
// (the shipped library is built without them); the two symbols verify.cpp takes from HIP translation units are restated here.
//   verify_fuzz <proof.bin> [iterations = 20000] [seed]
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <random>
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

int main(int argc, char** argv) {
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
