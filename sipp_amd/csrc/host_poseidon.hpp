// sipp_amd/csrc/host_poseidon.hpp -- host-side Poseidon-Goldilocks permutation (host_poseidon.cpp) of the Fiat-Shamir
// challenger: plain C++ / AVX-512, no HIP.  Same function as the device permutation of poseidon.hpp.
#pragma once
#include <stdint.h>

namespace host {
// in place; input words any u64 congruent to the state, output canonical
void poseidon_permute(uint64_t s[12]);
// a named implementation (0 = portable scalar, 1 = scalar with look-ahead partial rounds, 2 = AVX-512): 0 on success,
// -1 when the CPU lacks it, -2 for an unknown id.  For the tests that hold the implementations against each other.
int poseidon_permute_impl(uint64_t s[12], int impl);
}  // namespace host
