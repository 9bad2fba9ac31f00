// scripts/ubench/host_poseidon_bench.cpp -- latency of ONE dependent host permutation per implementation of
// sipp_amd/csrc/host_poseidon.cpp (0 scalar, 1 scalar + look-ahead, 2 AVX-512, 3 AVX-512 full rounds + scalar look-ahead)
// build: clang++ -O3 -std=c++17 -Isipp_amd/csrc scripts/ubench/host_poseidon_bench.cpp sipp_amd/csrc/host_poseidon.cpp -o /tmp/hpb
#include <chrono>
#include <cstdio>
#include <cstring>

#include "host_poseidon.hpp"
int main() {
    const int impls[] = {0, 1, 2, 3, 10, 11, 12, 13, 14};  // 10.. = pieces: full AVX-512, full scalar, partial scalar / look-ahead / AVX-512
    for (int impl : impls) {
        uint64_t s[12];
        for (int i = 0; i < 12; i++) s[i] = (uint64_t)i * 0x9E3779B97F4A7C15ULL % 0xFFFFFFFF00000001ULL;
        const int N = 400000;
        double best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; i++)
                if (host::poseidon_permute_impl(s, impl)) return 1;
            auto t1 = std::chrono::steady_clock::now();
            const double us = std::chrono::duration<double, std::micro>(t1 - t0).count() / N;
            if (us < best) best = us;
        }
        printf("impl %d: %.3f us per dependent permutation  (%016lx)\n", impl, best, (unsigned long)s[0]);
    }
    return 0;
}
