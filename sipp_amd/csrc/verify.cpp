// sipp_amd/csrc/verify.cpp -- the VERIFIER of this library's STARK proofs: sipp_stark_verify (include/sipp_hip.h).
//
// What it stands for in the reference: every `*_exp_circuit` proof generator of starky-bn254 calls starky's native
// `verify_stark_proof` on the proof it has just made (recalled), and `data.verify(proof)` of
// src/verifier_circuit.rs:254 rests on the same checks inside the recursive verifier.  Verification is HOST work by nature --
// about 40 k Poseidon permutations and one evaluation of the constraints at zeta, milliseconds on one core -- so this file is plain
// C++ (no HIP, no GPU needed): the proving side stays on the device, the check of a proof does not need one.
//
// Same protocol as the provers (stark.hip / fri.hip), read from the proof's own words:
//   header (16 words) | trace cap | Z cap | quotient cap | openings at zeta and g zeta | FRI (layer caps, final polynomial, PoW witness,
//   query rounds) | public inputs (the IO records, one u32 per word)
// The AIR is data (data/air_tables.h, the program both provers interpret); this is its evaluation over the quadratic extension.
// Nothing here comes from oracle/: the oracle's verifier (oracle/stark.c) is the independent reading the tests hold this one against,
// verdict by verdict and, for damaged proofs, refusal stage by refusal stage (tests/test_product_verifier.py).
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <thread>
#include <vector>

#include "../../include/sipp_hip.h"
#include "air_tables.h"
#include "gl.hpp"
#include "host_challenger.hpp"

const air_spec_t* sipp_air_get(int kind, uint32_t log_n);   // trace.hip: the AIR of an API kind at a trace length

namespace {

using gl::E2;
constexpr uint64_t MAGIC = 0x5349505053544b31ULL;   // "SIPPSTK1"

// ------------------------------------------------------------------------------------------------------------------------------
// BN254 base field on the host (4 x 64-bit limbs, Montgomery form): the PUBLIC side of a statement -- are the records' points on
// their curves, is the pairing's Q of order r, the sign rule of the map to G2, the tower limbs of an Fq12 value.
// ------------------------------------------------------------------------------------------------------------------------------
typedef unsigned __int128 u128;
struct Fq {
    uint64_t l[4];
};
const uint64_t BN_P[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
// r - 1 (the group order minus one): [r - 1] Q = -Q  <=>  [r] Q = O for Q != O
const uint64_t BN_RM1[4] = {0x43e1f593f0000000ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};

struct FqCtx {
    uint64_t ninv;   // -p^-1 mod 2^64
    Fq r2, one;      // 2^512 mod p, 2^256 mod p
};
bool fq_geq_p(const Fq& a) {
    for (int i = 3; i >= 0; i--) {
        if (a.l[i] > BN_P[i]) return true;
        if (a.l[i] < BN_P[i]) return false;
    }
    return true;
}
void fq_sub_p(Fq& a) {
    uint64_t bw = 0;
    for (int i = 0; i < 4; i++) {
        const u128 d = (u128)a.l[i] - BN_P[i] - bw;
        a.l[i] = (uint64_t)d;
        bw = (uint64_t)(d >> 64) & 1;
    }
}
Fq fq_add(const Fq& a, const Fq& b) {
    Fq r;
    uint64_t c = 0;
    for (int i = 0; i < 4; i++) {
        const u128 s = (u128)a.l[i] + b.l[i] + c;
        r.l[i] = (uint64_t)s;
        c = (uint64_t)(s >> 64);
    }
    if (c || fq_geq_p(r)) fq_sub_p(r);      // p < 2^254: the sum fits 255 bits, c is always 0
    return r;
}
Fq fq_sub(const Fq& a, const Fq& b) {
    Fq r;
    uint64_t bw = 0;
    for (int i = 0; i < 4; i++) {
        const u128 d = (u128)a.l[i] - b.l[i] - bw;
        r.l[i] = (uint64_t)d;
        bw = (uint64_t)(d >> 64) & 1;
    }
    if (bw) {
        uint64_t c = 0;
        for (int i = 0; i < 4; i++) {
            const u128 s = (u128)r.l[i] + BN_P[i] + c;
            r.l[i] = (uint64_t)s;
            c = (uint64_t)(s >> 64);
        }
    }
    return r;
}
bool fq_is_zero(const Fq& a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
bool fq_eq(const Fq& a, const Fq& b) { return a.l[0] == b.l[0] && a.l[1] == b.l[1] && a.l[2] == b.l[2] && a.l[3] == b.l[3]; }
// Montgomery product a b / 2^256 (coarsely integrated operand scanning)
Fq fq_mul(const FqCtx& k, const Fq& a, const Fq& b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a.l[j] * b.l[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (uint64_t)c;
        t[5] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * k.ninv;
        c = (u128)m * BN_P[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)m * BN_P[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = t[5] + (uint64_t)(c >> 64);
    }
    Fq r = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || fq_geq_p(r)) fq_sub_p(r);
    return r;
}
FqCtx fq_ctx() {
    FqCtx k;
    uint64_t x = 1;                       // Newton: x = p^-1 mod 2^64
    for (int i = 0; i < 6; i++) x *= 2 - BN_P[0] * x;
    k.ninv = 0 - x;
    Fq v = {{1, 0, 0, 0}};
    for (int i = 0; i < 256; i++) v = fq_add(v, v);
    k.one = v;
    for (int i = 0; i < 256; i++) v = fq_add(v, v);
    k.r2 = v;
    return k;
}
// eight u32 words (little-endian, canonical: checked before) -> Montgomery form
Fq fq_from_words(const FqCtx& k, const uint32_t* w) {
    Fq a;
    for (int i = 0; i < 4; i++) a.l[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
    return fq_mul(k, a, k.r2);
}
Fq fq_from_mont(const FqCtx& k, const Fq& a) {
    const Fq one = {{1, 0, 0, 0}};
    return fq_mul(k, a, one);
}
Fq fq_small(const FqCtx& k, uint64_t v) {
    const Fq a = {{v, 0, 0, 0}};
    return fq_mul(k, a, k.r2);
}
Fq fq_inv(const FqCtx& k, const Fq& a) {   // a^(p - 2)
    uint64_t e[4] = {BN_P[0] - 2, BN_P[1], BN_P[2], BN_P[3]};
    Fq r = k.one, b = a;
    for (int i = 0; i < 254; i++) {
        if ((e[i >> 6] >> (i & 63)) & 1) r = fq_mul(k, r, b);
        b = fq_mul(k, b, b);
    }
    return r;
}
bool words_canonical(const uint32_t* w) {
    for (int q = 3; q >= 0; q--) {
        const uint64_t v = (uint64_t)w[2 * q] | ((uint64_t)w[2 * q + 1] << 32);
        if (v < BN_P[q]) return true;
        if (v > BN_P[q]) return false;
    }
    return false;
}
// Fq2 = Fq[u] / (u^2 + 1)
struct Fq2 {
    Fq c0, c1;
};
Fq2 f2_add(const Fq2& a, const Fq2& b) { return Fq2{fq_add(a.c0, b.c0), fq_add(a.c1, b.c1)}; }
Fq2 f2_sub(const Fq2& a, const Fq2& b) { return Fq2{fq_sub(a.c0, b.c0), fq_sub(a.c1, b.c1)}; }
Fq2 f2_mul(const FqCtx& k, const Fq2& a, const Fq2& b) {
    return Fq2{fq_sub(fq_mul(k, a.c0, b.c0), fq_mul(k, a.c1, b.c1)), fq_add(fq_mul(k, a.c0, b.c1), fq_mul(k, a.c1, b.c0))};
}
bool f2_is_zero(const Fq2& a) { return fq_is_zero(a.c0) && fq_is_zero(a.c1); }
bool f2_eq(const Fq2& a, const Fq2& b) { return fq_eq(a.c0, b.c0) && fq_eq(a.c1, b.c1); }
Fq2 f2_inv(const FqCtx& k, const Fq2& a) {
    const Fq n = fq_inv(k, fq_add(fq_mul(k, a.c0, a.c0), fq_mul(k, a.c1, a.c1)));
    const Fq zero = {{0, 0, 0, 0}};
    return Fq2{fq_mul(k, a.c0, n), fq_sub(zero, fq_mul(k, a.c1, n))};
}
Fq2 f2_from_words(const FqCtx& k, const uint32_t* w) { return Fq2{fq_from_words(k, w), fq_from_words(k, w + 8)}; }
// y^2 = x^3 + 3 on E(Fq)
bool on_g1(const FqCtx& k, const Fq& x, const Fq& y) {
    return fq_eq(fq_sub(fq_mul(k, y, y), fq_mul(k, fq_mul(k, x, x), x)), fq_small(k, 3));
}
// (9 + u)(y^2 - x^3) = 3 on the twist E'(Fq2): y^2 = x^3 + 3 / (9 + u)
bool on_twist(const FqCtx& k, const Fq2& x, const Fq2& y) {
    const Fq2 d = f2_sub(f2_mul(k, y, y), f2_mul(k, f2_mul(k, x, x), x));
    const Fq nine = fq_small(k, 9);
    const Fq re = fq_sub(fq_mul(k, nine, d.c0), d.c1), im = fq_add(d.c0, fq_mul(k, nine, d.c1));
    return fq_eq(re, fq_small(k, 3)) && fq_is_zero(im);
}
// [r - 1] Q == -Q by affine double-and-add from the top bit (a step without an affine slope means Q is not of order r)
bool g2_order_r(const FqCtx& k, const Fq2& qx, const Fq2& qy) {
    Fq2 tx = qx, ty = qy;
    const Fq three = fq_small(k, 3);
    for (int i = 252; i >= 0; i--) {
        Fq2 den = f2_add(ty, ty);
        if (f2_is_zero(den)) return false;
        Fq2 t2 = f2_mul(k, tx, tx);
        t2 = Fq2{fq_mul(k, t2.c0, three), fq_mul(k, t2.c1, three)};
        Fq2 lam = f2_mul(k, t2, f2_inv(k, den));
        Fq2 x3 = f2_sub(f2_sub(f2_mul(k, lam, lam), tx), tx);
        ty = f2_sub(f2_mul(k, lam, f2_sub(tx, x3)), ty);
        tx = x3;
        if ((BN_RM1[i >> 6] >> (i & 63)) & 1) {
            den = f2_sub(qx, tx);
            if (f2_is_zero(den)) return false;
            lam = f2_mul(k, f2_sub(qy, ty), f2_inv(k, den));
            x3 = f2_sub(f2_sub(f2_mul(k, lam, lam), tx), qx);
            ty = f2_sub(f2_mul(k, lam, f2_sub(tx, x3)), ty);
            tx = x3;
        }
    }
    const Fq2 zero = Fq2{Fq{{0, 0, 0, 0}}, Fq{{0, 0, 0, 0}}};
    return f2_eq(tx, qx) && f2_eq(ty, f2_sub(zero, qy));
}
// sgn0 of RFC 9380 for Fq2 (the sign rule of the map to G2: src/bin/bls_aggregation.rs:102 through map_to_g2_without_cofactor_mul)
int f2_sgn0(const FqCtx& k, const Fq2& a) {
    const Fq c0 = fq_from_mont(k, a.c0), c1 = fq_from_mont(k, a.c1);
    const int s0 = (int)(c0.l[0] & 1), z0 = fq_is_zero(c0), s1 = (int)(c1.l[0] & 1);
    return s0 | (z0 & s1);
}

int base_kind(int kind) { return kind == SIPP_G1_EXP_HARDENED ? SIPP_G1_EXP : kind == SIPP_G2_EXP_HARDENED ? SIPP_G2_EXP : kind; }

// every Fq element of every record below p (the exponent of an exponentiation record may be any 256-bit value)
bool records_canonical(int kind, const uint32_t* pis, size_t num_io) {
    kind = base_kind(kind);
    if (kind == SIPP_MAP_G2 || kind == SIPP_PAIRING) {
        const size_t fe = kind == SIPP_MAP_G2 ? 6 : 18;
        for (size_t e = 0; e < fe * num_io; e++)
            if (!words_canonical(pis + 8 * e)) return false;
        return true;
    }
    const int fe = kind == SIPP_G1_EXP ? 2 : kind == SIPP_G2_EXP ? 4 : 12;
    const int ppi = 8 * (3 * fe + 1);
    for (size_t io = 0; io < num_io; io++)
        for (int e = 0; e < 3 * fe + 1; e++)
            if (e != 2 * fe && !words_canonical(pis + io * ppi + 8 * e)) return false;
    return true;
}
// the statement is about group elements (src/verifier_circuit.rs:92-124): x and offset of a G1 / G2 record on their curve, the point of
// a MapToG2 record on the twist with the map's sign, (P, Q) of a pairing record on E / E' with Q of order r
bool records_on_curve(const FqCtx& k, int kind, const uint32_t* pis, size_t num_io) {
    kind = base_kind(kind);
    for (size_t io = 0; io < num_io; io++) {
        if (kind == SIPP_PAIRING) {
            const uint32_t* rec = pis + io * SIPP_PAIRING_IO_WORDS;
            if (io > 0 && memcmp(rec, rec - SIPP_PAIRING_IO_WORDS, 48 * sizeof(uint32_t)) == 0) continue;   // a padding copy of the record before
            const Fq2 qx = f2_from_words(k, rec + 16), qy = f2_from_words(k, rec + 32);
            if (!on_g1(k, fq_from_words(k, rec), fq_from_words(k, rec + 8)) || !on_twist(k, qx, qy) || !g2_order_r(k, qx, qy)) return false;
        } else if (kind == SIPP_MAP_G2) {
            const uint32_t* rec = pis + io * SIPP_MAP_G2_IO_WORDS;
            const Fq2 u = f2_from_words(k, rec), x = f2_from_words(k, rec + 16), y = f2_from_words(k, rec + 32);
            if (!on_twist(k, x, y) || f2_sgn0(k, u) != f2_sgn0(k, y)) return false;
        } else if (kind == SIPP_G1_EXP) {
            const uint32_t* rec = pis + io * SIPP_G1_IO_WORDS;
            if (!on_g1(k, fq_from_words(k, rec), fq_from_words(k, rec + 8)) || !on_g1(k, fq_from_words(k, rec + 16), fq_from_words(k, rec + 24)))
                return false;
        } else if (kind == SIPP_G2_EXP) {
            const uint32_t* rec = pis + io * SIPP_G2_IO_WORDS;
            if (!on_twist(k, f2_from_words(k, rec), f2_from_words(k, rec + 16)) ||
                !on_twist(k, f2_from_words(k, rec + 32), f2_from_words(k, rec + 48)))
                return false;
        }
    }
    return true;
}

// ------------------------------------------------------------------------------------------------------------------------------
// Goldilocks helpers
// ------------------------------------------------------------------------------------------------------------------------------
E2 e2(uint64_t c0) { return E2{c0, 0}; }
E2 e2_from_i64(int64_t v) { return E2{gl::from_i64(v), 0}; }
// 1 / x for every entry (Montgomery's trick; no entry is zero)
void batch_inverse(std::vector<E2>& v) {
    std::vector<E2> pre(v.size());
    E2 acc = e2(1);
    for (size_t i = 0; i < v.size(); i++) {
        pre[i] = acc;
        acc = gl::mul(acc, v[i]);
    }
    E2 inv = gl::inv(acc);
    for (size_t i = v.size(); i-- > 0;) {
        const E2 x = v[i];
        v[i] = gl::mul(inv, pre[i]);
        inv = gl::mul(inv, x);
    }
}
// weights of the interpolation over the subgroup H of order 2^log_h at the point y outside H:
//   f(y) = sum_r f(h^r) c_r,  c_r = (y^|H| - 1) / |H| * h^r / (y - h^r)
// false when y lies in H
bool lagrange_weights(unsigned log_h, E2 y, std::vector<E2>& c) {
    const size_t H = (size_t)1 << log_h;
    const uint64_t h = gl::root_of_unity(log_h);
    c.resize(H);
    uint64_t hr = 1;
    for (size_t r = 0; r < H; r++) {
        c[r] = gl::sub(y, e2(hr));
        if (c[r].c0 == 0 && c[r].c1 == 0) return false;
        hr = gl::mul(hr, h);
    }
    batch_inverse(c);
    const E2 zh = gl::scale(gl::sub(gl::pow(y, (uint64_t)H), e2(1)), gl::inv((uint64_t)H % gl::P));
    hr = 1;
    for (size_t r = 0; r < H; r++) {
        c[r] = gl::mul(gl::scale(c[r], hr), zh);
        hr = gl::mul(hr, h);
    }
    return true;
}

struct Reader {
    const uint64_t* p;
    size_t pos, len;
    bool bad = false;
    const uint64_t* take(size_t n) {
        if (bad || n > len - pos) {
            bad = true;
            return p;       // (never dereferenced past a check of `bad`: callers test before use or read at most what a header has)
        }
        const uint64_t* r = p + pos;
        pos += n;
        return r;
    }
};

struct FriShape {
    uint32_t rate_bits, cap_height, pow_bits, num_queries, pow_rule;
    std::vector<uint32_t> arity_bits;
};
// FriReductionStrategy::ConstantArityBits(arity_bits, final_poly_bits)
FriShape fri_shape(const sipp_stark_config& c, unsigned degree_bits) {
    FriShape p;
    p.rate_bits = c.rate_bits; p.cap_height = c.cap_height; p.pow_bits = c.pow_bits; p.num_queries = c.num_queries; p.pow_rule = c.pow_rule;
    while (degree_bits > c.final_poly_bits && degree_bits + c.rate_bits - c.arity_bits >= c.cap_height && degree_bits >= c.arity_bits &&
           p.arity_bits.size() < 32) {
        p.arity_bits.push_back(c.arity_bits);
        degree_bits -= c.arity_bits;
    }
    return p;
}

// hash_or_noop(leaf) climbed to the cap entry the index points at
bool merkle_ok(const uint64_t* leaf, size_t leaf_len, size_t index, const uint64_t* siblings, size_t n_siblings, const uint64_t* cap) {
    uint64_t cur[4] = {0, 0, 0, 0};
    if (leaf_len <= 4) memcpy(cur, leaf, leaf_len * 8);
    else host::Challenger::hash_no_pad(leaf, leaf_len, cur);
    for (size_t l = 0; l < n_siblings; l++) {
        uint64_t nxt[4];
        if ((index >> l) & 1) host::Challenger::two_to_one(siblings + 4 * l, cur, nxt);
        else host::Challenger::two_to_one(cur, siblings + 4 * l, nxt);
        memcpy(cur, nxt, 32);
    }
    return memcmp(cur, cap + 4 * (index >> n_siblings), 32) == 0;
}

void observe_ext(host::Challenger& ch, E2 v) {
    ch.observe(v.c0);
    ch.observe(v.c1);
}

// ------------------------------------------------------------------------------------------------------------------------------
// the AIR program at one point of the extension field (constraint order = alpha-power order: acc <- acc alpha + c, as starky's
// ConstraintConsumer): gadgets (32 / group coefficient equations + the booleanity of the sign), polynomial constraints, then the range
// table (3), the permuted lookups (2 per checked column), the permutation products (2 per Z column)
// ------------------------------------------------------------------------------------------------------------------------------
struct Eval {
    const E2 *local, *next, *aux, *per, *z_local, *z_next;
    E2 lag_first, lag_last, z_last;
    uint64_t alpha[2], beta[2], gamma[2];
    E2 acc[2];
    void emit(E2 v) {
        acc[0] = gl::add(gl::scale(acc[0], alpha[0]), v);
        acc[1] = gl::add(gl::scale(acc[1], alpha[1]), v);
    }
    // a limb vector of the program: sum of terms coef [flag] cells(base + i stride); returns the words read
    size_t vec(const int64_t* w, E2* out, int& n_out) const {
        const int n = (int)w[0], nt = (int)w[1];
        for (int i = 0; i < n; i++) out[i] = e2(0);
        for (int t = 0; t < nt; t++) {
            const int64_t* tm = w + 2 + 5 * t;
            const int base = (int)tm[1], stride = (int)tm[2], flag = (int)tm[3], neg = (int)tm[4];
            E2 f = e2_from_i64(tm[0]);
            if (flag >= 0) {
                E2 pv = per[flag];
                if (neg) pv = gl::sub(e2(1), pv);
                f = gl::mul(f, pv);
            }
            for (int i = 0; i < n; i++) out[i] = gl::add(out[i], gl::mul(f, local[base + i * stride]));
        }
        n_out = n;
        return 2 + 5 * (size_t)nt;
    }
    void program(const air_spec_t* a) {
        const int64_t* w = a->prog;
        const int64_t* end = a->prog + a->prog_len;
        E2 va[17], vb[17], q[17], e[33];
        while (w < end) {
            if (w[0] == 1) {
                const int sign_col = (int)w[1], cbase = (int)w[2], ncl = (int)w[3], lb = (int)w[4];
                const int64_t coffset = w[5];
                const int grp = (int)w[6];
                w += 7;
                int nq, na, nb;
                w += vec(w, q, nq);
                for (int k = 0; k < 33; k++) e[k] = e2(0);
                const int np = (int)*w++;
                for (int p = 0; p < np; p++) {
                    const E2 coef = e2_from_i64(*w++);
                    w += vec(w, va, na);
                    bool live = false;
                    for (int i = 0; i < na; i++) live = live || va[i].c0 != 0 || va[i].c1 != 0;
                    if (!live) {              // (exactly zero operand: the product adds nothing)
                        w += 2 + 5 * (size_t)w[1];
                        continue;
                    }
                    w += vec(w, vb, nb);
                    for (int i = 0; i < na; i++) {
                        const E2 ai = gl::mul(coef, va[i]);
                        for (int j = 0; j < nb; j++) e[i + j] = gl::add(e[i + j], gl::mul(ai, vb[j]));
                    }
                }
                const int nl = (int)*w++;
                for (int p = 0; p < nl; p++) {
                    const E2 coef = e2_from_i64(*w++);
                    w += vec(w, va, na);
                    for (int i = 0; i < na; i++) e[i] = gl::add(e[i], gl::mul(coef, va[i]));
                }
                const E2 s = local[sign_col];
                const E2 sgn = gl::sub(e2(1), gl::add(s, s));
                E2 cprev = e2(0);
                const int nm = 32 / grp;
                for (int m = 0; m < nm; m++) {
                    E2 v = e2(0);
                    for (int t = grp - 1; t >= 0; t--) {
                        const int k = grp * m + t;
                        E2 qp = e2(0);
                        for (int i = 0; i < nq; i++) {
                            const int j = k - i;
                            if (j >= 0 && j < 16) qp = gl::add(qp, gl::scale(q[i], (uint64_t)AIR_BN_P_LIMBS[j]));
                        }
                        v = gl::add(gl::scale(v, 65536), gl::sub(e[k], gl::mul(sgn, qp)));
                    }
                    E2 ck = e2(0);
                    if (m < nm - 1) {
                        for (int l = 0; l < ncl; l++) ck = gl::add(ck, gl::scale(local[cbase + m * ncl + l], ((uint64_t)1 << (lb * l)) % gl::P));
                        ck = gl::sub(ck, e2_from_i64(coffset));
                    }
                    v = gl::sub(v, cprev);
                    v = gl::add(v, gl::scale(ck, ((uint64_t)1 << (16 * grp)) % gl::P));
                    emit(v);
                    cprev = ck;
                }
                emit(gl::mul(s, gl::sub(s, e2(1))));
            } else {
                const int nmono = (int)w[1];
                w += 2;
                E2 sum = e2(0);
                for (int m = 0; m < nmono; m++) {
                    E2 t = e2_from_i64(*w++);
                    const int nf = (int)*w++;
                    for (int f = 0; f < nf; f++) {
                        const int kind = (int)w[0], idx = (int)w[1];
                        w += 2;
                        t = gl::mul(t, kind == 0 ? local[idx] : kind == 1 ? next[idx] : kind == 2 ? aux[idx] : per[idx]);
                    }
                    sum = gl::add(sum, t);
                }
                emit(sum);
            }
        }
    }
    void all(const air_spec_t* a) {
        acc[0] = acc[1] = e2(0);
        program(a);
        const int nm = a->n_main, nc = a->n_checked;
        const E2 one = e2(1);
        // range table column 0: 0, 1, ..., T - 1, T - 1, ...
        const E2 tl = local[0], tn = next[0], d = gl::sub(tn, tl);
        emit(gl::mul(lag_first, tl));
        emit(gl::mul(z_last, gl::mul(d, gl::sub(d, one))));
        emit(gl::mul(lag_last, gl::sub(tl, e2_from_i64(((int64_t)1 << a->table_bits) - 1))));
        for (int j = 0; j < nc; j++) {
            const E2 pin = local[nm + j], ptab = local[nm + nc + j], npin = next[nm + j], nptab = next[nm + nc + j];
            emit(gl::mul(lag_first, gl::sub(pin, ptab)));
            emit(gl::mul(z_last, gl::mul(gl::sub(npin, pin), gl::sub(npin, nptab))));
        }
        // Z' (pin + gamma)(ptab + beta) = Z (col + gamma)(table + beta), one Z per checked column per challenge
        for (int i = 0; i < 2; i++) {
            const E2 g = e2(gamma[i]), b = e2(beta[i]);
            for (int j = 0; j < nc; j++) {
                const E2 z = z_local[i * nc + j], zn = z_next[i * nc + j];
                const E2 lhs = gl::mul(gl::add(local[a->checked_base + j], g), gl::add(tl, b));
                const E2 rhs = gl::mul(gl::add(local[nm + j], g), gl::add(local[nm + nc + j], b));
                emit(gl::mul(lag_first, gl::sub(z, one)));
                emit(gl::sub(gl::mul(zn, rhs), gl::mul(z, lhs)));
            }
        }
    }
};

// ------------------------------------------------------------------------------------------------------------------------------
// FRI verification (plonky2 fri/verifier.rs as recalled: verify_fri_proof with PrecomputedReducedOpenings, fri_combine_initial,
// compute_evaluation per layer, the final polynomial) over any number of oracles and opening batches.  The reader stands behind the
// opened values; `opened[b]` are batch b's values in opening order; alpha has been drawn.  Query rounds are independent once their
// indices are drawn and have one size -- the oracle rows with their paths + every layer's coset with its path -- so a few threads check
// them; the verdict is the FIRST refusing round's, as if they had been read one after the other.
// ------------------------------------------------------------------------------------------------------------------------------
struct FriBatchV {
    E2 point;
    uint32_t n_ranges;
    const sipp_poly_range* ranges;
};
size_t batch_len(const FriBatchV& b) {
    size_t k = 0;
    for (uint32_t r = 0; r < b.n_ranges; r++) k += b.ranges[r].col_end - b.ranges[r].col_begin;
    return k;
}
int fri_verify(Reader& rb, const uint64_t* const* caps, const uint32_t* ncols, const uint32_t* n_salt, size_t n_oracles, const FriBatchV* batches,
               const E2* const* opened, size_t n_batches, unsigned log_n, const FriShape& fp, E2 alpha, host::Challenger& ch) {
    const uint64_t* proof = rb.p;
    const size_t len = rb.len;
    const unsigned log_m = log_n + fp.rate_bits;
    const size_t m = (size_t)1 << log_m, n = (size_t)1 << log_n;
    const size_t cap_n = (size_t)1 << fp.cap_height;
    if (fp.cap_height > log_m) return 120;
    std::vector<const uint64_t*> rcaps(fp.arity_bits.size());
    std::vector<E2> betas(fp.arity_bits.size());
    unsigned sum_ab = 0;
    for (size_t r = 0; r < fp.arity_bits.size(); r++) {
        rcaps[r] = rb.take(cap_n * 4);
        if (rb.bad) return 120;
        ch.observe_many(rcaps[r], cap_n * 4);
        betas[r] = ch.get_ext();
        if (fp.arity_bits[r] < 1 || fp.arity_bits[r] > 4) return 120;
        sum_ab += fp.arity_bits[r];
    }
    if (sum_ab > log_n) return 120;
    const size_t flen = n >> sum_ab;
    std::vector<E2> fpoly(flen ? flen : 1);
    {
        const uint64_t* e = rb.take(2 * flen);
        if (rb.bad) return 120;
        for (size_t i = 0; i < flen; i++) {
            fpoly[i] = E2{e[2 * i], e[2 * i + 1]};
            observe_ext(ch, fpoly[i]);
        }
    }
    const uint64_t* pwp = rb.take(1);
    if (rb.bad) return 120;
    uint64_t resp;
    if (fp.pow_rule == SIPP_POW_HASH) {
        uint64_t in[5], out[4];
        for (int i = 0; i < 4; i++) in[i] = ch.get();
        in[4] = *pwp;
        host::Challenger::hash_no_pad(in, 5, out);
        resp = out[0];
    } else {
        ch.observe(*pwp);
        resp = ch.get();
    }
    if (fp.pow_bits && (resp >> (64 - fp.pow_bits)) != 0) return 121;
    // reduced openings per batch: sum_j alpha^j opened_j, and alpha^(values of the batch)
    std::vector<E2> red(n_batches), shf(n_batches);
    for (size_t b = 0; b < n_batches; b++) {
        const size_t k = batch_len(batches[b]);
        E2 acc = e2(0);
        for (size_t j = k; j-- > 0;) acc = gl::add(gl::mul(acc, alpha), opened[b][j]);
        red[b] = acc;
        shf[b] = gl::pow(alpha, (uint64_t)k);
    }
    const uint64_t wm = gl::root_of_unity(log_m);
    const unsigned ns0 = log_m - fp.cap_height;
    size_t qwords = 0;
    for (size_t o = 0; o < n_oracles; o++) qwords += (size_t)ncols[o] + (size_t)n_salt[o] + (size_t)ns0 * 4;
    {
        unsigned lt = log_m;
        for (size_t r = 0; r < fp.arity_bits.size(); r++) {
            lt -= fp.arity_bits[r];
            qwords += ((size_t)2 << fp.arity_bits[r]) + (size_t)(lt > fp.cap_height ? lt - fp.cap_height : 0) * 4;
        }
    }
    std::vector<size_t> xs(fp.num_queries);
    for (uint32_t qi = 0; qi < fp.num_queries; qi++) xs[qi] = (size_t)(ch.get() % m);
    const size_t q_base = rb.pos;
    auto query = [&](uint32_t qi) -> int {
        Reader rq{proof, q_base + (size_t)qi * qwords, len};
        if (rq.pos > len) return 122;
        const size_t x = xs[qi];
        std::vector<const uint64_t*> rows(n_oracles);
        for (size_t o = 0; o < n_oracles; o++) {
            const size_t ll = (size_t)ncols[o] + (size_t)n_salt[o];
            rows[o] = rq.take(ll);
            const uint64_t* sib = rq.take((size_t)ns0 * 4);
            if (rq.bad) return 122;
            if (!merkle_ok(rows[o], ll, x, sib, ns0, caps[o])) return 123 + (int)(o < 3 ? o : 3);
        }
        uint64_t sub_x = gl::mul(gl::GEN, gl::pow(wm, (uint64_t)gl::bitrev((uint32_t)x, log_m)));
        // fri_combine_initial (salt words are never combined)
        E2 sum = e2(0);
        for (size_t b = 0; b < n_batches; b++) {
            E2 acc = e2(0), ap = e2(1);
            for (uint32_t r = 0; r < batches[b].n_ranges; r++) {
                const sipp_poly_range& rg = batches[b].ranges[r];
                for (uint32_t c = rg.col_begin; c < rg.col_end; c++) {
                    acc = gl::add(acc, gl::scale(ap, rows[rg.oracle][c]));
                    ap = gl::mul(ap, alpha);
                }
            }
            const E2 num = gl::sub(acc, red[b]), den = gl::sub(e2(sub_x), batches[b].point);
            if (den.c0 == 0 && den.c1 == 0) return 122;
            sum = gl::add(gl::mul(sum, shf[b]), gl::mul(num, gl::inv(den)));
        }
        E2 old = gl::scale(sum, sub_x);        // the final polynomial was multiplied by X
        size_t xi = x;
        unsigned log_tree = log_m;
        for (size_t r = 0; r < fp.arity_bits.size(); r++) {
            const unsigned ab = fp.arity_bits[r];
            const size_t arity = (size_t)1 << ab;
            log_tree -= ab;
            const unsigned ns = log_tree > fp.cap_height ? log_tree - fp.cap_height : 0;
            const uint64_t* evw = rq.take(2 * arity);
            const uint64_t* sib = rq.take((size_t)ns * 4);
            if (rq.bad) return 130;
            const size_t within = xi & (arity - 1);
            E2 evs[16];
            for (size_t k = 0; k < arity; k++) evs[k] = E2{evw[2 * k], evw[2 * k + 1]};
            if (!gl::eq(evs[within], old)) return 131;
            // compute_evaluation: interpolate {(coset_start w^i, evs[bitrev(i)])} and evaluate at beta_r
            const uint64_t w = gl::root_of_unity(ab);
            const uint64_t rev_within = gl::bitrev((uint32_t)within, ab);
            const uint64_t coset_start = gl::mul(sub_x, gl::pow(w, (uint64_t)arity - rev_within));
            uint64_t pts[16];
            for (size_t i = 0; i < arity; i++) pts[i] = gl::mul(coset_start, gl::pow(w, (uint64_t)i));
            E2 acc = e2(0);
            for (size_t i = 0; i < arity; i++) {
                E2 numr = evs[gl::bitrev((uint32_t)i, ab)];
                uint64_t den = 1;
                for (size_t k = 0; k < arity; k++) {
                    if (k == i) continue;
                    numr = gl::mul(numr, gl::sub(betas[r], e2(pts[k])));
                    den = gl::mul(den, gl::sub(pts[i], pts[k]));
                }
                acc = gl::add(acc, gl::scale(numr, gl::inv(den)));
            }
            old = acc;
            xi >>= ab;
            if (!merkle_ok(evw, 2 * arity, xi, sib, ns, rcaps[r])) return 132;
            sub_x = gl::pow(sub_x, (uint64_t)arity);
        }
        E2 fv = e2(0);
        for (size_t i = flen; i-- > 0;) fv = gl::add(gl::scale(fv, sub_x), fpoly[i]);
        if (!gl::eq(fv, old)) return 133;
        return 0;
    };
    std::vector<int> verdict(fp.num_queries, 0);
    {
        unsigned nt = std::thread::hardware_concurrency();
        nt = nt == 0 ? 1 : nt > 8 ? 8 : nt;
        if (nt > fp.num_queries) nt = fp.num_queries;
        auto work = [&](unsigned t) {
            for (uint32_t qi = t; qi < fp.num_queries; qi += nt) verdict[qi] = query(qi);
        };
        std::vector<std::thread> pool;
        unsigned started = 1;
        try {
            for (; started < nt; started++) pool.emplace_back(work, started);
        } catch (...) {                    // no more threads to be had: the rounds nobody took are checked here
            for (uint32_t qi = 0; qi < fp.num_queries; qi++)
                if (qi % nt >= started) verdict[qi] = query(qi);
        }
        if (nt) work(0);
        for (std::thread& th : pool) th.join();
    }
    for (uint32_t qi = 0; qi < fp.num_queries; qi++)
        if (verdict[qi]) return verdict[qi];
    rb.pos = q_base + (size_t)fp.num_queries * qwords;
    return 0;
}

// the 16-bit tower-basis limb `sub` (component sub / 16, limb sub % 16) of the MyFq12 value at rec[word .. word + 96):
// component 2 i = c_i + 9 c_(i+6), component 2 i + 1 = c_(i+6)
uint64_t tower_limb(const FqCtx& k, const uint32_t* rec, int sub) {
    const int t = sub / 16, i = t >> 1;
    Fq v = fq_from_words(k, rec + 8 * (i + 6));
    if ((t & 1) == 0) v = fq_add(fq_from_words(k, rec + 8 * i), fq_mul(k, fq_small(k, 9), v));
    v = fq_from_mont(k, v);
    const int l = sub % 16;
    return (v.l[l >> 2] >> (16 * (l & 3))) & 0xffff;
}
uint64_t aux_value(const FqCtx& k, const air_spec_t* a, const uint32_t* pis, size_t io, int ai) {
    const int word = a->aux[4 * ai], part = a->aux[4 * ai + 1], sub = a->aux[4 * ai + 3];
    const uint32_t* rec = pis + io * (size_t)a->pi_per_io;
    if (part == 3) return tower_limb(k, rec + word, sub);
    const uint32_t w = rec[word];
    return part == 0 ? (w & 0xffffu) : part == 1 ? (w >> 16) : w;
}

int log_rows_of(int kind) { return kind == SIPP_MAP_G2 ? 3 : kind == SIPP_PAIRING ? AIR_PAIRING_LOG_ROWS : 9; }

int verify(const uint64_t* proof, size_t len, const sipp_stark_config& cfg) {
    Reader rb{proof, 0, len};
    if (len < 16) return 100;
    const uint64_t* h = rb.take(16);
    if (h[0] != MAGIC) return 100;
    for (int i = 1; i < 16; i++)
        if (h[i] >> 32) return 100;          // every header word is a small integer: no high bits to hide a second encoding in
    // every body word is a field element in canonical form (x + p would hash and compute like x: a second encoding of the same proof)
    for (size_t i = 16; i < len; i++)
        if (proof[i] >= gl::P) return 141;
    const int kind = (int)h[1];
    const unsigned log_n = (unsigned)h[2];
    const size_t num_io = (size_t)h[3];
    const int W = (int)h[4], P = (int)h[5], Q = (int)h[6];
    if (kind < 0 || kind > SIPP_PAIRING || log_n < 10 || log_n > 26) return 101;
    const unsigned log_rows = (unsigned)log_rows_of(kind);
    if (log_n <= log_rows || num_io != ((size_t)1 << (log_n - log_rows))) return 101;       // at least two records' blocks
    const air_spec_t* a = sipp_air_get(kind, log_n);
    if (cfg.fs_rule > 1 || cfg.lookup_rule > 1 || cfg.rate_bits < 1 || cfg.rate_bits > 3 || cfg.cap_height > 8 || cfg.arity_bits < 1 ||
        cfg.arity_bits > 4 || cfg.num_queries < 1 || cfg.num_queries > 1024 || cfg.pow_bits > 32 || cfg.pow_rule > 1)
        return 102;
    if (!a || (unsigned)a->log_rows != log_rows || W != a->n_main + 2 * a->n_checked || P != 2 * a->n_checked || Q != 4 ||
        h[7] != cfg.cap_height || h[10] != cfg.num_queries || (int)h[11] != a->pi_per_io || h[12] != len || h[13] != cfg.rate_bits ||
        h[14] != cfg.arity_bits || h[15] != (uint64_t)(cfg.fs_rule | (cfg.lookup_rule << 1)))
        return 102;
    const size_t n = (size_t)1 << log_n;
    const FriShape fp = fri_shape(cfg, log_n);
    if (h[8] != fp.arity_bits.size()) return 103;
    const size_t cap_n = (size_t)1 << cfg.cap_height;
    const size_t n_pi = num_io * (size_t)a->pi_per_io;
    if (len < 16 + n_pi) return 104;
    std::vector<uint32_t> pis(n_pi ? n_pi : 1);
    for (size_t k = 0; k < n_pi; k++) {
        const uint64_t v = proof[len - n_pi + k];
        if (v >> 32) return 105;
        pis[k] = (uint32_t)v;
    }
    if (!records_canonical(kind, pis.data(), num_io)) return 108;
    const FqCtx fk = fq_ctx();
    if (!records_on_curve(fk, kind, pis.data(), num_io)) return 109;

    host::Challenger ch;
    if (cfg.fs_rule == SIPP_FS_STATEMENT) {
        // the statement ahead of the trace cap: kind, shape, configuration, Merkle root of the records
        const uint64_t st[16] = {(uint64_t)kind, log_n, num_io, (uint64_t)W, (uint64_t)P, (uint64_t)Q, cfg.rate_bits, cfg.cap_height,
                                 cfg.pow_bits, cfg.arity_bits, cfg.final_poly_bits, cfg.num_queries, cfg.num_challenges,
                                 cfg.pow_rule, (uint64_t)a->pi_per_io, cfg.lookup_rule};
        ch.observe_many(st, 16);
        const size_t ppi = (size_t)a->pi_per_io;
        std::vector<uint64_t> d(num_io * 4), tmp(ppi);
        for (size_t io = 0; io < num_io; io++) {
            for (size_t k = 0; k < ppi; k++) tmp[k] = pis[io * ppi + k];
            host::Challenger::hash_no_pad(tmp.data(), ppi, &d[4 * io]);
        }
        for (size_t cnt = num_io; cnt > 1; cnt >>= 1)
            for (size_t i = 0; i < cnt / 2; i++) {
                uint64_t out[4];
                host::Challenger::two_to_one(&d[8 * i], &d[8 * i + 4], out);
                memcpy(&d[4 * i], out, 32);
            }
        ch.observe_many(d.data(), 4);
    }
    const uint64_t* trace_cap = rb.take(cap_n * 4);
    if (rb.bad) return 106;
    ch.observe_many(trace_cap, cap_n * 4);
    uint64_t beta[2], gamma[2], alpha[2];
    for (int i = 0; i < 2; i++) {
        beta[i] = ch.get();
        gamma[i] = ch.get();
        if (cfg.lookup_rule) beta[i] = gamma[i];
    }
    const uint64_t* z_cap = rb.take(cap_n * 4);
    if (rb.bad) return 106;
    ch.observe_many(z_cap, cap_n * 4);
    alpha[0] = ch.get();
    alpha[1] = ch.get();
    const uint64_t* q_cap = rb.take(cap_n * 4);
    if (rb.bad) return 106;
    ch.observe_many(q_cap, cap_n * 4);
    const E2 zeta = ch.get_ext();
    const size_t n_open = (size_t)(2 * W + 2 * P + Q);
    const uint64_t* opw = rb.take(2 * n_open);
    if (rb.bad) return 106;
    std::vector<E2> op(n_open);
    for (size_t k = 0; k < n_open; k++) op[k] = E2{opw[2 * k], opw[2 * k + 1]};
    // opened values in the order the prover observed them: local | Z | quotient at zeta, then next | Z next at g zeta
    for (int c = 0; c < W; c++) observe_ext(ch, op[c]);
    for (int c = 0; c < P; c++) observe_ext(ch, op[2 * W + c]);
    for (int c = 0; c < Q; c++) observe_ext(ch, op[2 * W + 2 * P + c]);
    for (int c = 0; c < W; c++) observe_ext(ch, op[W + c]);
    for (int c = 0; c < P; c++) observe_ext(ch, op[2 * W + P + c]);

    // ---- the constraints at zeta ----
    const uint64_t g = gl::root_of_unity(log_n), gi = gl::inv(g);
    const E2 zn = gl::pow(zeta, (uint64_t)n);
    const E2 zh = gl::sub(zn, e2(1));
    if (zh.c0 == 0 && zh.c1 == 0) return 107;
    {
        Eval ev;
        const int n_vper = a->n_vflag + a->n_vconst;
        std::vector<E2> per((size_t)AIR_N_PERIODIC + (size_t)n_vper);
        const uint64_t ninv = gl::inv((uint64_t)n % gl::P);
        // closed-form selectors S_(m, r0)(x) = (K / N) ((x g^-r0)^N - 1) / ((x g^-r0)^K - 1), K = N / m: 1 on the rows r = r0 mod m
        for (int k = 0; k < AIR_N_PERIODIC; k++) {
            const uint64_t m = (uint64_t)AIR_PERIODIC[k][0], r0 = (uint64_t)AIR_PERIODIC[k][1], K = (uint64_t)n / m;
            const E2 y = gl::scale(zeta, gl::inv(gl::pow(g, r0)));
            const E2 num = gl::sub(gl::pow(y, (uint64_t)n), e2(1)), den = gl::sub(gl::pow(y, K), e2(1));
            if (den.c0 == 0 && den.c1 == 0) return 107;
            per[k] = gl::scale(gl::mul(num, gl::inv(den)), gl::mul(K % gl::P, ninv));
        }
        // the AIR's value-periodic columns (the pairing AIR's selectors and row constants): column k is P_k(x^(N / R)), P_k interpolating
        // its R values over the subgroup of order R -- evaluated through the interpolation weights at y = zeta^(N / R), one set for all
        if (n_vper > 0) {
            const unsigned lr = (unsigned)a->log_rows;
            const size_t R = (size_t)1 << lr;
            std::vector<E2> cw;
            if (!lagrange_weights(lr, gl::pow(zeta, (uint64_t)1 << (log_n - lr)), cw)) return 107;
            for (int kk = 0; kk < n_vper; kk++) {
                E2 acc = e2(0);
                for (size_t r = 0; r < R; r++) {
                    const int64_t v = air_vper_value(a, kk, (int)r);
                    if (v == 0) continue;
                    acc = v == 1 ? gl::add(acc, cw[r]) : v == -1 ? gl::sub(acc, cw[r]) : gl::add(acc, gl::scale(cw[r], gl::from_i64(v)));
                }
                per[(size_t)AIR_N_PERIODIC + (size_t)kk] = acc;
            }
        }
        // public-input polynomials: A_ai(g^(rows io + shift)) = value(io), i.e. the interpolation over the subgroup of order num_io at
        // zeta g^-shift; one set of weights per distinct shift
        std::vector<E2> auxz((size_t)(a->n_aux ? a->n_aux : 1));
        {
            const unsigned log_io = log_n - log_rows;
            std::vector<int> shifts;
            for (int ai = 0; ai < a->n_aux; ai++) {
                const int sh = a->aux[4 * ai + 2];
                bool seen = false;
                for (int s : shifts) seen = seen || s == sh;
                if (!seen) shifts.push_back(sh);
            }
            std::vector<E2> cw;
            for (int sh : shifts) {
                // A(x) = I(x g^-shift), I (degree < num_io) interpolating the values over the subgroup of order num_io (g^rows generates it)
                const E2 y = gl::scale(zeta, gl::inv(gl::pow(g, (uint64_t)sh)));
                if (!lagrange_weights(log_io, y, cw)) return 107;
                for (int ai = 0; ai < a->n_aux; ai++) {
                    if (a->aux[4 * ai + 2] != sh) continue;
                    E2 acc = e2(0);
                    for (size_t io = 0; io < num_io; io++) acc = gl::add(acc, gl::scale(cw[io], aux_value(fk, a, pis.data(), io, ai)));
                    auxz[(size_t)ai] = acc;
                }
            }
        }
        ev.local = op.data();
        ev.next = op.data() + W;
        ev.aux = auxz.data();
        ev.per = per.data();
        ev.z_local = op.data() + 2 * W;
        ev.z_next = op.data() + 2 * W + P;
        // L_first, L_last, x - g^-1 (zero on the last row)
        const E2 zhn = gl::scale(zh, ninv);
        const E2 d_first = gl::sub(zeta, e2(1)), d_last = gl::sub(zeta, e2(gi));
        if ((d_first.c0 == 0 && d_first.c1 == 0) || (d_last.c0 == 0 && d_last.c1 == 0)) return 107;
        ev.lag_first = gl::mul(zhn, gl::inv(d_first));
        ev.lag_last = gl::mul(gl::scale(zhn, gi), gl::inv(d_last));
        ev.z_last = d_last;
        for (int i = 0; i < 2; i++) {
            ev.alpha[i] = alpha[i];
            ev.beta[i] = beta[i];
            ev.gamma[i] = gamma[i];
        }
        ev.all(a);
        for (int i = 0; i < 2; i++) {
            const E2 qz = gl::add(op[2 * W + 2 * P + 2 * i], gl::mul(zn, op[2 * W + 2 * P + 2 * i + 1]));
            if (!gl::eq(ev.acc[i], gl::mul(zh, qz))) return 110 + i;
        }
    }

    // ---- FRI: the batch (local | Z | quotient) opened at zeta, (local | Z) at g zeta ----
    const E2 fa = ch.get_ext();
    if ((size_t)h[9] != (n >> (fp.arity_bits.size() * cfg.arity_bits))) return 120;
    const sipp_poly_range r0[3] = {{0, 0, (uint32_t)W}, {1, 0, (uint32_t)P}, {2, 0, (uint32_t)Q}};
    const FriBatchV batches[2] = {{zeta, 3, r0}, {gl::scale(zeta, g), 2, r0}};
    std::vector<E2> o0((size_t)(W + P + Q)), o1((size_t)(W + P));
    for (int c = 0; c < W; c++) { o0[c] = op[c]; o1[c] = op[W + c]; }
    for (int c = 0; c < P; c++) { o0[W + c] = op[2 * W + c]; o1[W + c] = op[2 * W + P + c]; }
    for (int c = 0; c < Q; c++) o0[W + P + c] = op[2 * W + 2 * P + c];
    const E2* opened[2] = {o0.data(), o1.data()};
    const uint64_t* caps3[3] = {trace_cap, z_cap, q_cap};
    const uint32_t ncols3[3] = {(uint32_t)W, (uint32_t)P, (uint32_t)Q}, salt3[3] = {0, 0, 0};
    const int rc = fri_verify(rb, caps3, ncols3, salt3, 3, batches, opened, 2, log_n, fp, fa, ch);
    if (rc) return rc;
    if (rb.pos + n_pi != len) return 140;
    return 0;
}


// ------------------------------------------------------------------------------------------------------------------------------
// PolynomialBatch opening proofs ("SIPPFRI1", sipp_fri_prove_openings) and the outer prover's proofs ("SIPPPLK3", sipp_plonk_prove_gates)
// ------------------------------------------------------------------------------------------------------------------------------
constexpr uint64_t FRI_MAGIC = 0x5349505046524931ULL;    // "SIPPFRI1"
constexpr uint64_t PLONK_MAGIC3 = 0x334b4c5050504953ULL;  // "SIPPPLK3" (as bytes in memory)

FriShape fri_shape_of(const sipp_fri_params& p) {
    FriShape f;
    f.rate_bits = p.rate_bits; f.cap_height = p.cap_height; f.pow_bits = p.pow_bits; f.num_queries = p.num_queries; f.pow_rule = p.pow_rule;
    for (uint32_t r = 0; r < p.n_rounds && r < SIPP_FRI_MAX_ROUNDS; r++) f.arity_bits.push_back(p.arity_bits[r]);
    return f;
}
bool fri_params_ok(const sipp_fri_params& p, unsigned log_n) {
    return p.rate_bits >= 1 && p.rate_bits <= 8 && p.cap_height <= 16 && p.cap_height <= log_n + p.rate_bits && p.pow_bits <= 32 && p.num_queries >= 1 &&
           p.num_queries <= 1024 && p.pow_rule <= 1 && p.n_rounds <= SIPP_FRI_MAX_ROUNDS && log_n >= 1 && log_n <= 26;
}

// the section written by PolynomialBatch::prove_openings: header[8], the opened values batch by batch (observed as they are read), then
// what fri_verify reads.  `sec` / `sec_len`: the section alone (its header carries its own length)
int fri_openings_verify(const uint64_t* sec, size_t sec_len, const uint64_t* const* caps, const uint32_t* ncols, const uint32_t* n_salt,
                        size_t n_oracles, const FriBatchV* batches, size_t n_batches, unsigned log_n, const sipp_fri_params& p, host::Challenger& ch) {
    if (sec_len < 8 || sec[0] != FRI_MAGIC || sec[1] != p.n_rounds || sec[3] != p.num_queries || sec[4] != n_oracles || sec[5] != n_batches ||
        sec[6] != sec_len || sec[7] != log_n)
        return 100;
    {   // header word 2 = length of the final polynomial: fixed by the parameters (an unchecked word would be a second encoding of the proof)
        unsigned sum_ab = 0;
        for (uint32_t r = 0; r < p.n_rounds && r < SIPP_FRI_MAX_ROUNDS; r++) sum_ab += p.arity_bits[r];
        if (sum_ab > log_n || sec[2] != (((uint64_t)1 << log_n) >> sum_ab)) return 100;
    }
    for (size_t i = 8; i < sec_len; i++)
        if (sec[i] >= gl::P) return 141;
    for (size_t b = 0; b < n_batches; b++)
        for (uint32_t r = 0; r < batches[b].n_ranges; r++) {
            const sipp_poly_range& rg = batches[b].ranges[r];
            if (rg.oracle >= n_oracles || rg.col_begin > rg.col_end || rg.col_end > ncols[rg.oracle]) return 100;
        }
    Reader rb{sec, 8, sec_len};
    std::vector<std::vector<E2>> opened(n_batches);
    std::vector<const E2*> optr(n_batches);
    for (size_t b = 0; b < n_batches; b++) {
        const size_t k = batch_len(batches[b]);
        const uint64_t* e = rb.take(2 * k);
        if (rb.bad) return 106;
        opened[b].resize(k ? k : 1);
        for (size_t j = 0; j < k; j++) {
            opened[b][j] = E2{e[2 * j], e[2 * j + 1]};
            observe_ext(ch, opened[b][j]);
        }
        optr[b] = opened[b].data();
    }
    const E2 alpha = ch.get_ext();
    const int rc = fri_verify(rb, caps, ncols, n_salt, n_oracles, batches, optr.data(), n_batches, log_n, fri_shape_of(p), alpha, ch);
    if (rc) return rc;
    return rb.pos == sec_len ? 0 : 140;
}

// a gate set handed over as data: operands in range, programs inside program_words (what sipp_plonk_prove_gates refuses too)
bool circuit_ok(const sipp_plonk_circuit& c, const sipp_plonk_params& p) {
    if (c.num_wires < p.num_routed_wires || c.num_selectors == 0 || c.num_selectors > c.num_constants || c.num_gates == 0 || !c.gates ||
        (!c.programs && c.program_words))
        return false;
    for (uint32_t g = 0; g < c.num_gates; g++) {
        const sipp_plonk_gate& ga = c.gates[g];
        if (ga.selector_index >= c.num_selectors || ga.group_lo > ga.row || ga.row >= ga.group_hi || ga.group_hi > c.num_gates) return false;
        size_t w = ga.prog_offset;
        for (uint32_t j = 0; j < ga.num_constraints; j++) {
            if (w >= c.program_words) return false;
            const int64_t nm = c.programs[w++];
            if (nm < 0 || nm > 4096) return false;
            for (int64_t mo = 0; mo < nm; mo++) {
                if (w + 2 > c.program_words) return false;
                const int64_t nf = c.programs[w + 1];
                w += 2;
                if (nf < 0 || nf > 64 || w + 2 * (size_t)nf > c.program_words) return false;
                for (int64_t f = 0; f < nf; f++, w += 2) {
                    const int64_t kind = c.programs[w], idx = c.programs[w + 1];
                    if (kind < 0 || kind > 2 || idx < 0 || (kind == 0 && idx >= c.num_wires) || (kind == 1 && idx >= c.num_constants) ||
                        (kind == 2 && idx >= 4))
                        return false;
                }
            }
        }
    }
    return true;
}

// plonk/verifier.rs with the gate constraints evaluated at zeta from the OPENED constants and wires (gates as data: per gate a selector
// filter and one polynomial per constraint), the permutation argument's Z(1) = 1 and partial-product terms, one reduce_with_powers per
// challenge against the quotient chunks, then the opening proof
int plonk_verify(const uint64_t* proof, size_t len, const uint64_t* cs_cap, const sipp_plonk_params& p, const sipp_fri_params& fp,
                 const sipp_plonk_circuit& c, const uint64_t digest[4]) {
    if (p.num_routed_wires == 0 || p.max_degree < 2 || p.max_degree > 64 || (p.max_degree & (p.max_degree - 1)) || p.num_challenges == 0 ||
        p.num_challenges > 8 || !circuit_ok(c, p))
        return 201;
    const uint32_t R = p.num_routed_wires, D = p.max_degree, C = p.num_challenges, np = (R + D - 1) / D - 1, nz = C * (1 + np), m = np + 1;
    const uint32_t W = c.num_wires, K = c.num_constants;
    uint32_t ngc = 0;
    for (uint32_t g = 0; g < c.num_gates; g++) ngc = c.gates[g].num_constraints > ngc ? c.gates[g].num_constraints : ngc;
    if (fp.cap_height > 16) return 201;
    const size_t cap_n = (size_t)1 << fp.cap_height;
    if (len < 16 + 3 * cap_n * 4 || proof[0] != PLONK_MAGIC3 || proof[2] != R || proof[3] != D || proof[4] != C || proof[5] != len || proof[6] != W ||
        proof[7] != K || proof[8] != c.num_selectors || proof[9] != c.num_gates || proof[10] != ngc || (proof[12] | proof[13] | proof[14] | proof[15]))
        return 201;
    const size_t n_pi = (size_t)proof[11];
    if (proof[11] > len - (16 + 3 * cap_n * 4)) return 201;
    const unsigned log_n = (unsigned)proof[1];
    if (proof[1] < 1 || proof[1] > 26 || !fri_params_ok(fp, log_n)) return 202;
    // every body word is a field element in canonical form (x + p would hash and compute like x: a second encoding of the same proof)
    for (size_t i = 16; i < len; i++)
        if (proof[i] >= gl::P) return 141;
    uint64_t pih[4];
    host::Challenger::hash_no_pad(proof + len - n_pi, n_pi, pih);
    const uint64_t *wcap = proof + 16, *zcap = wcap + cap_n * 4, *qcap = zcap + cap_n * 4, *op = qcap + cap_n * 4;
    const size_t op_len = len - (size_t)(op - proof) - n_pi;
    host::Challenger ch;
    ch.observe_many(digest, 4);
    ch.observe_many(pih, 4);
    ch.observe_many(wcap, cap_n * 4);
    uint64_t betas[8], gammas[8], alphas[8];
    for (uint32_t i = 0; i < C; i++) betas[i] = ch.get();
    for (uint32_t i = 0; i < C; i++) gammas[i] = ch.get();
    ch.observe_many(zcap, cap_n * 4);
    for (uint32_t i = 0; i < C; i++) alphas[i] = ch.get();
    ch.observe_many(qcap, cap_n * 4);
    const E2 zeta = ch.get_ext();
    const size_t n0 = (size_t)K + R + W + nz + (size_t)C * D, n_open = n0 + C;
    if (op_len < 8 + 2 * n_open) return 203;
    std::vector<E2> v(n_open);
    for (size_t k = 0; k < n_open; k++) v[k] = E2{op[8 + 2 * k], op[8 + 2 * k + 1]};
    const E2 *cv = v.data(), *sg = cv + K, *wv = cv + K + R, *zs = wv + W, *pps = zs + C, *qs = zs + nz, *zs_next = cv + n0;
    // gate constraint terms: term j = sum over gates of filter_g(selector) * constraint_(g, j)
    std::vector<E2> terms;
    const uint64_t n = (uint64_t)1 << log_n;
    const E2 zeta_n = gl::pow(zeta, n), zh = gl::sub(zeta_n, e2(1));
    {
        const E2 d1 = gl::sub(zeta, e2(1));
        if (d1.c0 == 0 && d1.c1 == 0) return 107;
        const E2 l0 = gl::mul(zh, gl::inv(gl::scale(d1, n % gl::P)));
        for (uint32_t cc = 0; cc < C; cc++) terms.push_back(gl::mul(l0, gl::sub(zs[cc], e2(1))));
        for (uint32_t cc = 0; cc < C; cc++)
            for (uint32_t q = 0; q < m; q++) {
                E2 num = e2(1), den = e2(1);
                for (uint32_t j = q * D; j < (q + 1) * D && j < R; j++) {
                    const E2 sid = gl::scale(zeta, gl::mul(betas[cc], gl::pow(7, (uint64_t)j)));       // k_j = 7^j: get_unique_coset_shifts
                    num = gl::mul(num, gl::add(gl::add(wv[j], sid), e2(gammas[cc])));
                    den = gl::mul(den, gl::add(gl::add(wv[j], gl::scale(sg[j], betas[cc])), e2(gammas[cc])));
                }
                const E2 prev = q == 0 ? zs[cc] : pps[cc * np + q - 1], next = q == np ? zs_next[cc] : pps[cc * np + q];
                terms.push_back(gl::sub(gl::mul(prev, num), gl::mul(next, den)));
            }
        std::vector<E2> gt(ngc ? ngc : 1, e2(0));
        for (uint32_t g = 0; g < c.num_gates; g++) {
            const sipp_plonk_gate& ga = c.gates[g];
            E2 f = e2(1);
            const E2 sel = cv[ga.selector_index];
            for (uint32_t i = ga.group_lo; i < ga.group_hi; i++)
                if (i != ga.row) f = gl::mul(f, gl::sub(e2(i), sel));
            if (c.num_selectors > 1) f = gl::mul(f, gl::sub(e2(0xffffffffull), sel));
            const int64_t* w = c.programs + ga.prog_offset;
            for (uint32_t j = 0; j < ga.num_constraints; j++) {
                const int64_t nm = *w++;
                E2 sum = e2(0);
                for (int64_t mo = 0; mo < nm; mo++) {
                    E2 t = e2_from_i64(w[0]);
                    const int64_t nf = w[1];
                    w += 2;
                    for (int64_t k = 0; k < nf; k++, w += 2) t = gl::mul(t, w[0] == 0 ? wv[w[1]] : w[0] == 1 ? cv[w[1]] : e2(pih[w[1]]));
                    sum = gl::add(sum, t);
                }
                gt[j] = gl::add(gt[j], gl::mul(f, sum));
            }
        }
        for (uint32_t j = 0; j < ngc; j++) terms.push_back(gt[j]);
    }
    for (uint32_t cc = 0; cc < C; cc++) {
        E2 van = e2(0);
        for (size_t k = terms.size(); k-- > 0;) van = gl::add(gl::scale(van, alphas[cc]), terms[k]);
        E2 acc = e2(0);
        for (uint32_t d = D; d-- > 0;) acc = gl::add(gl::mul(acc, zeta_n), qs[cc * D + d]);
        if (!gl::eq(van, gl::mul(zh, acc))) return 210;
    }
    const uint64_t* caps[4] = {cs_cap, wcap, zcap, qcap};
    const uint32_t ncols[4] = {K + R, W, nz, C * D}, salt[4] = {0, 0, 0, 0};
    const sipp_poly_range r0[4] = {{0, 0, K + R}, {1, 0, W}, {2, 0, nz}, {3, 0, C * D}}, r1[1] = {{2, 0, C}};
    const FriBatchV batches[2] = {{zeta, 4, r0}, {gl::scale(zeta, gl::root_of_unity(log_n)), 1, r1}};
    return fri_openings_verify(op, op_len, caps, ncols, salt, 4, batches, 2, log_n, fp, ch);
}

}  // namespace

extern "C" int sipp_fri_verify_openings(const uint64_t* proof, size_t len, const uint64_t* const* caps, const uint32_t* ncols, const uint32_t* n_salt,
                                        size_t n_oracles, const sipp_fri_batch* batches, size_t n_batches, uint32_t log_n, const sipp_fri_params* p,
                                        sipp_challenger* chal, int* reason) {
    if (reason) *reason = 0;
    if (!proof || !caps || !ncols || !batches || !p || !chal || n_oracles == 0 || n_oracles > 64 || n_batches == 0 || n_batches > 64) return SIPP_E_BADARG;
    if (!fri_params_ok(*p, log_n)) return SIPP_E_UNSUPPORTED;
    std::vector<FriBatchV> bv(n_batches);
    for (size_t b = 0; b < n_batches; b++) bv[b] = FriBatchV{E2{batches[b].point[0], batches[b].point[1]}, batches[b].n_ranges, batches[b].ranges};
    std::vector<uint32_t> zero(n_oracles, 0);
    host::Challenger ch;
    memcpy(ch.state, chal->state, sizeof ch.state);
    memcpy(ch.in_buf, chal->in_buf, sizeof ch.in_buf);
    memcpy(ch.out_buf, chal->out_buf, sizeof ch.out_buf);
    if (chal->n_in > 8 || chal->n_out > 8) return SIPP_E_BADARG;
    ch.n_in = (uint32_t)chal->n_in;
    ch.n_out = (uint32_t)chal->n_out;
    int r;
    try {
        r = fri_openings_verify(proof, len, caps, ncols, n_salt ? n_salt : zero.data(), n_oracles, bv.data(), n_batches, log_n, *p, ch);
    } catch (...) {                        // (allocation failure: nothing crosses the C boundary)
        return SIPP_E_NOMEM;
    }
    memcpy(chal->state, ch.state, sizeof ch.state);
    memcpy(chal->in_buf, ch.in_buf, sizeof ch.in_buf);
    memcpy(chal->out_buf, ch.out_buf, sizeof ch.out_buf);
    chal->n_in = ch.n_in;
    chal->n_out = ch.n_out;
    if (reason) *reason = r;
    return r == 0 ? SIPP_OK : SIPP_E_VERIFY;
}

extern "C" int sipp_plonk_verify_gates(const uint64_t* proof, size_t len, const uint64_t* constants_sigmas_cap, const sipp_plonk_params* p,
                                       const sipp_fri_params* fp, const sipp_plonk_circuit* c, const uint64_t circuit_digest[4], int* reason) {
    if (reason) *reason = 0;
    if (!proof || !constants_sigmas_cap || !p || !fp || !c || !circuit_digest) return SIPP_E_BADARG;
    int r;
    try {
        r = plonk_verify(proof, len, constants_sigmas_cap, *p, *fp, *c, circuit_digest);
    } catch (...) {
        return SIPP_E_NOMEM;
    }
    if (reason) *reason = r;
    return r == 0 ? SIPP_OK : SIPP_E_VERIFY;
}

extern "C" int sipp_stark_verify(const uint64_t* proof, size_t len, const sipp_stark_config* cfg, int* reason) {
    if (reason) *reason = 0;
    if (!proof) return SIPP_E_BADARG;
    sipp_stark_config c;
    if (cfg) c = *cfg;
    else sipp_default_config(&c);
    if (c.num_challenges != 2) return SIPP_E_UNSUPPORTED;
    int r;
    try {
        r = verify(proof, len, c);
    } catch (...) {
        return SIPP_E_NOMEM;
    }
    if (reason) *reason = r;
    return r == 0 ? SIPP_OK : SIPP_E_VERIFY;
}
