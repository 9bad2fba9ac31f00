// sipp_amd/csrc/host_challenger.hpp -- the host-side Fiat-Shamir challenger (plonky2 iop/challenger.rs as recalled)
// over the host Poseidon permutation: plain C++, no HIP.  Used by the provers (stark.hip, fri.hip, plonk.hip) and by the host-only
// verifier (verify.cpp).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "gl.hpp"
#include "host_poseidon.hpp"

namespace host {

// The permutation itself lives in host_poseidon.cpp (portable scalar and AVX-512 forms, bit-identical, chosen at load time):
// observing the ~32 k opening words of the widest STARK is ~4 k sequential permutations on the proof's critical path.
// Checked against the device kernel by every proof parity test and against the CPU oracle by tests/test_abi.py.

struct Challenger {
    uint64_t state[12] = {0};
    uint64_t in_buf[8];
    uint32_t n_in = 0;
    uint64_t out_buf[8];
    uint32_t n_out = 0;
    void duplex() {
        for (uint32_t i = 0; i < n_in; i++) state[i] = in_buf[i];
        n_in = 0;
        poseidon_permute(state);
        for (int i = 0; i < 8; i++) out_buf[i] = state[i];
        n_out = 8;
    }
    void observe(uint64_t e) {
        n_out = 0;
        in_buf[n_in++] = e;
        if (n_in == 8) duplex();
    }
    void observe_many(const uint64_t* e, size_t n) {
        for (size_t i = 0; i < n; i++) observe(e[i]);
    }
    uint64_t get() {
        if (n_in != 0 || n_out == 0) duplex();
        return out_buf[--n_out];
    }
    // hash_n_to_hash_no_pad (overwrite-mode sponge, rate 8) and two_to_one, for the statement binding of stark.hip
    static void hash_no_pad(const uint64_t* in, size_t n, uint64_t out[4]) {
        uint64_t s[12] = {0};
        for (size_t i = 0; i < n; i += 8) {
            for (size_t k = 0; k < 8 && i + k < n; k++) s[k] = in[i + k];
            poseidon_permute(s);
        }
        for (int k = 0; k < 4; k++) out[k] = s[k];
    }
    static void two_to_one(const uint64_t l[4], const uint64_t r[4], uint64_t out[4]) {
        uint64_t s[12] = {l[0], l[1], l[2], l[3], r[0], r[1], r[2], r[3], 0, 0, 0, 0};
        poseidon_permute(s);
        for (int k = 0; k < 4; k++) out[k] = s[k];
    }
    gl::E2 get_ext() {
        gl::E2 r;
        r.c0 = get();
        r.c1 = get();
        return r;
    }
};

}  // namespace host
