// sipp_amd/csrc/poseidon_quad.hpp -- Poseidon-Goldilocks with FOUR lanes per state (16 states per wave).
//
// Why: a sponge over W columns is sequential per leaf, so a launch has only (#leaves / 64) waves when every
// lane owns a whole state.  The Fq12 STARK (2^14 leaves x ~1,700 permutations) and the upper Merkle levels
// then leave three quarters of the SIMDs idle.  Splitting a state over the 4 lanes of a DPP quad gives 4x
// the waves at ~75 % lane efficiency -- the cross-lane traffic is v_mov_b32 with quad_perm (full-rate VALU,
// no LDS crossbar, no memory).
//
// Layout: lane q of a quad holds elements 3q, 3q+1, 3q+2.  Because 3 * 4 = 12, rotating by k lanes inside
// the quad yields the elements at offset +3k (mod 12) in EVERY lane, so the circulant MDS uses the same
// coefficient C[(3k + j - j') mod 12] in all lanes (only the DIAG[0] term is lane dependent).
// Per-element constants (round constants, sparse-layer vectors) differ per lane and come from an LDS copy
// of the tables.  Bit-exact with the one-state-per-lane permutation (tests/test_gpu_generic.py).
#pragma once
#include "poseidon.hpp"

namespace poseidon_quad {

// LDS table layout (u64 words)
constexpr int T_RC = 0;            // 360
constexpr int T_FIRST = 360;       // 12
constexpr int T_SCALAR = 372;      // 22
constexpr int T_MI = 394;          // 121
constexpr int T_VS = 515;          // 242
constexpr int T_WHAT = 757;        // 242
constexpr int T_WORDS = 999;

__device__ __forceinline__ void load_tables(uint64_t* tab) {
    for (int i = threadIdx.x; i < T_WORDS; i += blockDim.x) {
        uint64_t v;
        if (i < T_FIRST) v = poseidon::c_rc[i];
        else if (i < T_SCALAR) v = poseidon::c_fast_first[i - T_FIRST];
        else if (i < T_MI) v = poseidon::c_fast_scalar[i - T_SCALAR];
        else if (i < T_VS) v = poseidon::c_fast_mi[i - T_MI];
        else if (i < T_WHAT) v = poseidon::c_fast_vs[i - T_VS];
        else v = poseidon::c_fast_what[i - T_WHAT];
        tab[i] = v;
    }
    __syncthreads();
}

// value held by lane (q + K) % 4 of the same quad
template <int K>
__device__ __forceinline__ uint32_t quad_rot32(uint32_t v) {
    constexpr int ctrl = ((0 + K) & 3) | (((1 + K) & 3) << 2) | (((2 + K) & 3) << 4) | (((3 + K) & 3) << 6);
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, ctrl, 0xf, 0xf, true);
}
template <int K>
__device__ __forceinline__ uint64_t quad_rot(uint64_t v) {
    return ((uint64_t)quad_rot32<K>((uint32_t)(v >> 32)) << 32) | quad_rot32<K>((uint32_t)v);
}
// broadcast lane 0 of the quad
__device__ __forceinline__ uint64_t quad_bcast0(uint64_t v) {
    uint32_t lo = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)v, 0, 0xf, 0xf, true);
    uint32_t hi = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)(v >> 32), 0, 0xf, 0xf, true);
    return ((uint64_t)hi << 32) | lo;
}

struct Gathered {
    uint32_t lo[12], hi[12];  // index 3k + j : element at offset +3k+j from this lane's first element
};

__device__ __forceinline__ void gather(const uint64_t s[3], Gathered& g) {
#pragma unroll
    for (int j = 0; j < 3; j++) {
        uint32_t l = (uint32_t)s[j], h = (uint32_t)(s[j] >> 32);
        g.lo[j] = l;
        g.hi[j] = h;
        g.lo[3 + j] = quad_rot32<1>(l);
        g.hi[3 + j] = quad_rot32<1>(h);
        g.lo[6 + j] = quad_rot32<2>(l);
        g.hi[6 + j] = quad_rot32<2>(h);
        g.lo[9 + j] = quad_rot32<3>(l);
        g.hi[9 + j] = quad_rot32<3>(h);
    }
}

// circulant MDS on the quad: out[3q + j'] = sum_{d=0..11} C[(d - j') mod 12] * in[(3q + d) mod 12] (+ 8 in[0] for element 0)
__device__ __forceinline__ void mds_full(uint64_t s[3], uint32_t diag0 /* 8 in lane 0 of the quad, else 0 */) {
    constexpr uint32_t CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    Gathered g;
    gather(s, g);
#pragma unroll
    for (int jp = 0; jp < 3; jp++) {
        uint64_t al = 0, ah = 0;
#pragma unroll
        for (int d = 0; d < 12; d++) {
            al += (uint64_t)g.lo[d] * CIRC[(d - jp + 12) % 12];
            ah += (uint64_t)g.hi[d] * CIRC[(d - jp + 12) % 12];
        }
        if (jp == 0) {
            al += (uint64_t)g.lo[0] * diag0;
            ah += (uint64_t)g.hi[0] * diag0;
        }
        uint64_t l = al + (ah << 32);
        uint32_t h = (uint32_t)(ah >> 32) + (l < al ? 1u : 0u);
        s[jp] = gl::reduce96_nc(h, l);
    }
}

__device__ __forceinline__ uint64_t full_value(const Gathered& g, int d) { return ((uint64_t)g.hi[d] << 32) | g.lo[d]; }

// s: this lane's three elements (3q + j); q = lane & 3; tab = LDS tables
__device__ __forceinline__ void permute(uint64_t s[3], const uint32_t q, const uint64_t* tab) {
    const uint32_t diag0 = q == 0 ? 8u : 0u;
    const uint32_t e0 = 3 * q;
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int j = 0; j < 3; j++) s[j] = poseidon::sbox_thin(gl::add_nc(s[j], tab[T_RC + 12 * r + e0 + j]));
        mds_full(s, diag0);
    }
    // ---- partial rounds, sparse form ----
#pragma unroll
    for (int j = 0; j < 3; j++) s[j] = gl::add_nc(s[j], tab[T_FIRST + e0 + j]);
    {
        // dense pre-multiplication of elements 1..11 (element 0 passes through)
        Gathered g;
        gather(s, g);
        uint64_t t[3];
#pragma unroll
        for (int jp = 0; jp < 3; jp++) {
            const uint32_t e = e0 + jp;              // output element
            const uint32_t row = e == 0 ? 0 : e - 1; // row of the 11x11 block (lane 0, jp 0 is discarded)
            poseidon::Acc160 acc;
#pragma unroll
            for (int d = 0; d < 12; d++) {
                // input element (e0 + d) mod 12; skip element 0 (coefficient 0)
                const uint32_t ie = (e0 + d) % 12;
                const uint64_t cv = tab[T_MI + row * 11 + (ie ? ie - 1 : 0)];
                acc.mac(full_value(g, d), ie ? cv : 0);
            }
            t[jp] = acc.reduce();
        }
        if (q != 0) s[0] = t[0];
        s[1] = t[1];
        s[2] = t[2];
    }
#pragma unroll 1
    for (int r = 0; r < 22; r++) {
        // x = sbox(element 0) + scalar, computed by every lane on its s[0], taken from lane 0
        uint64_t x = gl::add_nc(poseidon::sbox_thin(s[0]), tab[T_SCALAR + r]);
        x = quad_bcast0(x);
        poseidon::Acc160 acc;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const uint32_t e = e0 + j;
            const uint64_t wv = tab[T_WHAT + r * 11 + (e ? e - 1 : 0)];
            acc.mac(e ? s[j] : x, e ? wv : 25);  // element 0 contributes x * M00
        }
        uint64_t part = gl::canon(acc.reduce());
        // sum over the quad (canonical modular adds)
        part = gl::add(part, quad_rot<1>(part));
        part = gl::add(part, quad_rot<2>(part));
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const uint32_t e = e0 + j;
            const uint64_t v = tab[T_VS + r * 11 + (e ? e - 1 : 0)];
            const uint64_t upd = gl::mad_nc(x, v, s[j]);
            s[j] = e ? upd : part;
        }
    }
#pragma unroll 1
    for (int r = 26; r < 30; r++) {
#pragma unroll
        for (int j = 0; j < 3; j++) s[j] = poseidon::sbox_thin(gl::add_nc(s[j], tab[T_RC + 12 * r + e0 + j]));
        mds_full(s, diag0);
    }
#pragma unroll
    for (int j = 0; j < 3; j++) s[j] = gl::canon(s[j]);
}

}  // namespace poseidon_quad
