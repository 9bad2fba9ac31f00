// sipp_amd/csrc/native.hip -- the native SIPP prover / verifier chain on top of the device routines (host code only).
//
// SURVEY.md section 8(f) rank 3 (and rows a14-a16 in product form): what the reference does on one CPU thread before the
// circuit is proved --
//   Transcript                reference src/transcript_native.rs:14-77   (hash_n_to_hash_no_pad over u32 limbs; get_challenge
//                                                                         with BigUint::to_u32_digits' dropped zero limbs)
//   sipp_prove_native         reference src/prover_native.rs:26-80
//   sipp_verify_native        reference src/verifier_native.rs:14-85
//   SIPPStatement layout      reference src/statements.rs:24-39, 134-169
//   obligation lists          reference src/verifier_circuit.rs:68-131   (the IO records of the three STARKs)
// with the pairings (sipp_inner_products, pairing.hip) and the folds A' = A1 + [x] A2, B' = B1 + [1/x] B2,
// Z' = Z Z_L^x Z_R^(1/x) (sipp_exp_outputs, the trace kernels' accumulator chains) on the GPU.  The transcript and the
// scalar-field arithmetic (a few hundred Poseidon permutations and one inversion mod r per round) stay on the host.
#include <vector>

#include "ctx.hpp"
#include "prover.hpp"

int sipp_inner_products_groups(sipp_ctx* ctx, const uint32_t* g1, const uint32_t* g2, const uint32_t* off, size_t count,
                               uint32_t* out);  // pairing.hip

namespace {

// ---- hash_n_to_hash_no_pad: overwrite-mode sponge, rate 8 ----
void hash_no_pad(const std::vector<uint64_t>& in, uint64_t out[4]) {
    uint64_t s[12] = {0};
    for (size_t off = 0; off < in.size(); off += 8) {
        const size_t m = std::min<size_t>(8, in.size() - off);
        for (size_t i = 0; i < m; i++) s[i] = in[off + i];
        host::poseidon_permute(s);
    }
    for (int i = 0; i < 4; i++) out[i] = s[i];
}

// ---- 256-bit integers mod r (the BN254 scalar field), little-endian u64 limbs; only a handful of operations per round ----
struct U256 {
    uint64_t w[4];
};
const U256 FR = {{0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull}};

int cmp(const U256& a, const U256& b) {
    for (int i = 3; i >= 0; i--)
        if (a.w[i] != b.w[i]) return a.w[i] < b.w[i] ? -1 : 1;
    return 0;
}
uint64_t sub_in_place(U256& a, const U256& b) {
    unsigned __int128 bw = 0;
    for (int i = 0; i < 4; i++) {
        const unsigned __int128 d = (unsigned __int128)a.w[i] - b.w[i] - bw;
        a.w[i] = (uint64_t)d;
        bw = (d >> 64) & 1;
    }
    return (uint64_t)bw;
}
const U256 FQ = {{0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull}};  // the base field

// ---- Montgomery arithmetic mod an odd 254-bit modulus (r or p), four 64-bit limbs, CIOS ----
struct Mont {
    U256 m;
    uint64_t ninv;  // -m^-1 mod 2^64
    U256 r2;        // 2^512 mod m
    explicit Mont(const U256& mod) : m(mod) {
        uint64_t x = mod.w[0];  // Newton: x <- x (2 - m x), correct to 3 bits at the start for odd m
        for (int i = 0; i < 6; i++) x *= 2 - mod.w[0] * x;
        ninv = (uint64_t)0 - x;
        // 2^512 mod m by 512 modular doublings of 1
        U256 t = {{1, 0, 0, 0}};
        for (int i = 0; i < 512; i++) {
            uint64_t c = 0;
            for (int k = 0; k < 4; k++) {
                const uint64_t nc = t.w[k] >> 63;
                t.w[k] = (t.w[k] << 1) | c;
                c = nc;
            }
            if (cmp(t, m) >= 0) sub_in_place(t, m);
        }
        r2 = t;
    }
    // a b 2^-256 mod m (a, b < m)
    U256 mul(const U256& a, const U256& b) const {
        uint64_t t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4; i++) {
            unsigned __int128 c = 0;
            for (int j = 0; j < 4; j++) {
                c += (unsigned __int128)a.w[j] * b.w[i] + t[j];
                t[j] = (uint64_t)c;
                c >>= 64;
            }
            c += t[4];
            t[4] = (uint64_t)c;
            t[5] = (uint64_t)(c >> 64);
            const uint64_t q = t[0] * ninv;
            c = (unsigned __int128)q * m.w[0] + t[0];
            c >>= 64;
            for (int j = 1; j < 4; j++) {
                c += (unsigned __int128)q * m.w[j] + t[j];
                t[j - 1] = (uint64_t)c;
                c >>= 64;
            }
            c += t[4];
            t[3] = (uint64_t)c;
            t[4] = t[5] + (uint64_t)(c >> 64);
        }
        U256 r = {{t[0], t[1], t[2], t[3]}};
        if (t[4] || cmp(r, m) >= 0) sub_in_place(r, m);
        return r;
    }
    U256 to(const U256& a) const { return mul(a, r2); }
    U256 from(const U256& a) const { return mul(a, U256{{1, 0, 0, 0}}); }
};
const Mont MONT_R(FR), MONT_P(FQ);

U256 inv_mod_r(const U256& x) {  // x^(r - 2)
    U256 e = FR;
    const U256 two = {{2, 0, 0, 0}};
    sub_in_place(e, two);
    const U256 xm = MONT_R.to(x);
    U256 r = MONT_R.to(U256{{1, 0, 0, 0}});
    for (int i = 253; i >= 0; i--) {
        r = MONT_R.mul(r, r);
        if ((e.w[i >> 6] >> (i & 63)) & 1) r = MONT_R.mul(r, xm);
    }
    return MONT_R.from(r);
}
void to_u32(const U256& a, uint32_t out[8]) {
    for (int i = 0; i < 4; i++) {
        out[2 * i] = (uint32_t)a.w[i];
        out[2 * i + 1] = (uint32_t)(a.w[i] >> 32);
    }
}

U256 from_u32(const uint32_t* w) {
    U256 a;
    for (int i = 0; i < 4; i++) a.w[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
    return a;
}
U256 addmod(const U256& a, const U256& b, const U256& m) {
    U256 s;
    unsigned __int128 cy = 0;
    for (int k = 0; k < 4; k++) {
        cy += (unsigned __int128)a.w[k] + b.w[k];
        s.w[k] = (uint64_t)cy;
        cy >>= 64;
    }
    if (cmp(s, m) >= 0) sub_in_place(s, m);
    return s;
}
U256 submod(const U256& a, const U256& b, const U256& m) {
    U256 d = a;
    if (sub_in_place(d, b)) {  // borrowed: add m back
        unsigned __int128 cy = 0;
        for (int k = 0; k < 4; k++) {
            cy += (unsigned __int128)d.w[k] + m.w[k];
            d.w[k] = (uint64_t)cy;
            cy >>= 64;
        }
    }
    return d;
}
// product in the reference's MyFq12 form Fq[w]/(w^12 - 18 w^6 + 82), 12 x 8 u32 limbs: the few products the verifier needs
// to chain the per-round powers (everything heavier runs on the device)
void f12_mul_host(const uint32_t* a, const uint32_t* b, uint32_t* out) {
    U256 x[12], y[12], d[23];
    for (int i = 0; i < 12; i++) {
        x[i] = from_u32(a + 8 * i);
        y[i] = from_u32(b + 8 * i);
        while (cmp(x[i], FQ) >= 0) sub_in_place(x[i], FQ);  // a tampered message may carry non-canonical limbs
        while (cmp(y[i], FQ) >= 0) sub_in_place(y[i], FQ);
        x[i] = MONT_P.to(x[i]);  // x 2^256, y 2^256: the Montgomery products below then give (x y) 2^256 ...
        y[i] = MONT_P.to(y[i]);
    }
    for (auto& v : d) v = U256{{0, 0, 0, 0}};
    for (int i = 0; i < 12; i++)
        for (int j = 0; j < 12; j++) d[i + j] = addmod(d[i + j], MONT_P.mul(x[i], y[j]), FQ);
    const U256 c18 = MONT_P.to(U256{{18, 0, 0, 0}}), c82 = MONT_P.to(U256{{82, 0, 0, 0}});
    for (int m = 22; m >= 12; m--) {
        d[m - 6] = addmod(d[m - 6], MONT_P.mul(d[m], c18), FQ);
        d[m - 12] = submod(d[m - 12], MONT_P.mul(d[m], c82), FQ);
    }
    for (int i = 0; i < 12; i++) to_u32(MONT_P.from(d[i]), out + 8 * i);  // ... and this takes the 2^256 off again
}

struct Transcript {
    uint64_t state[4] = {0, 0, 0, 0};
    // state <- H(state || msg), every u32 limb one field element (transcript_native.rs:23-30, 68-77)
    void append(const uint32_t* msg, size_t words) {
        std::vector<uint64_t> v(4 + words);
        for (int i = 0; i < 4; i++) v[i] = state[i];
        for (size_t i = 0; i < words; i++) v[4 + i] = msg[i];
        hash_no_pad(v, state);
    }
    // transcript_native.rs:56-65: the digest's u64 words as base-2^32 digits, where to_u32_digits() DROPS high zero limbs of
    // every word (and yields nothing for a zero word) before the concatenation; then reduced mod r
    U256 get_challenge() const {
        uint64_t d[4];
        hash_no_pad(std::vector<uint64_t>(state, state + 4), d);
        uint32_t digits[8];
        int nd = 0;
        for (int i = 0; i < 4; i++) {
            if (d[i] == 0) continue;
            digits[nd++] = (uint32_t)d[i];
            if (d[i] >> 32) digits[nd++] = (uint32_t)(d[i] >> 32);
        }
        U256 b = {{0, 0, 0, 0}};
        for (int i = 0; i < nd; i++) b.w[i >> 1] |= (uint64_t)digits[i] << (32 * (i & 1));
        while (cmp(b, FR) >= 0) sub_in_place(b, FR);
        return b;
    }
};

constexpr size_t G1W = 16, G2W = 32, F12W = 96;

bool is_pow2(size_t n) { return n && !(n & (n - 1)); }

// the obligation records of one round: {x: A2, offset: A1, exp_val: x} (verifier_circuit.rs:92-96), {x: B2, offset: B1, exp_val: 1/x};
// x == nullptr leaves the exponent words zero (sipp_fold_begin does not read them)
void fold_records(const std::vector<uint32_t>& A, const std::vector<uint32_t>& B, size_t n, const uint32_t* x, const uint32_t* ix,
                  std::vector<uint32_t>& rec1, std::vector<uint32_t>& rec2) {
    const size_t h = n / 2;
    rec1.assign(h * SIPP_G1_IO_WORDS, 0);
    rec2.assign(h * SIPP_G2_IO_WORDS, 0);
    for (size_t i = 0; i < h; i++) {
        uint32_t* r = &rec1[i * SIPP_G1_IO_WORDS];
        memcpy(r, &A[(h + i) * G1W], G1W * 4);
        memcpy(r + 16, &A[i * G1W], G1W * 4);
        if (x) memcpy(r + 32, x, 32);
        uint32_t* s = &rec2[i * SIPP_G2_IO_WORDS];
        memcpy(s, &B[(h + i) * G2W], G2W * 4);
        memcpy(s + 32, &B[i * G2W], G2W * 4);
        if (ix) memcpy(s + 64, ix, 32);
    }
}

// one round's folds on the device: A' = A1 + [x] A2 (G1 obligations {x: A2, offset: A1, exp_val: x}, verifier_circuit.rs:92-96)
// and B' = B1 + [1/x] B2; the complete IO records (outputs included) are left in rec1 / rec2
int fold_round(sipp_ctx* ctx, const std::vector<uint32_t>& A, const std::vector<uint32_t>& B, size_t n, const uint32_t x[8],
               const uint32_t ix[8], std::vector<uint32_t>& rec1, std::vector<uint32_t>& rec2) {
    const size_t h = n / 2;
    rec1.assign(h * SIPP_G1_IO_WORDS, 0);
    rec2.assign(h * SIPP_G2_IO_WORDS, 0);
    for (size_t i = 0; i < h; i++) {
        uint32_t* r = &rec1[i * SIPP_G1_IO_WORDS];
        memcpy(r, &A[(h + i) * G1W], G1W * 4);
        memcpy(r + 16, &A[i * G1W], G1W * 4);
        memcpy(r + 32, x, 32);
        uint32_t* s = &rec2[i * SIPP_G2_IO_WORDS];
        memcpy(s, &B[(h + i) * G2W], G2W * 4);
        memcpy(s + 32, &B[i * G2W], G2W * 4);
        memcpy(s + 64, ix, 32);
    }
    return sipp_fold_outputs(ctx, rec1.data(), h, rec2.data(), h);
}

void take_outputs(const std::vector<uint32_t>& rec1, const std::vector<uint32_t>& rec2, size_t h, std::vector<uint32_t>& A,
                  std::vector<uint32_t>& B) {
    A.resize(h * G1W);
    B.resize(h * G2W);
    for (size_t i = 0; i < h; i++) {
        memcpy(&A[i * G1W], &rec1[i * SIPP_G1_IO_WORDS + 40], G1W * 4);
        memcpy(&B[i * G2W], &rec2[i * SIPP_G2_IO_WORDS + 72], G2W * 4);
    }
}

// Z_L = <A2, B1>, Z_R = <A1, B2> in one device pass
int cross_products(sipp_ctx* ctx, const std::vector<uint32_t>& A, const std::vector<uint32_t>& B, size_t n, uint32_t zl[96],
                   uint32_t zr[96]) {
    const size_t h = n / 2;
    std::vector<uint32_t> g1(n * G1W), g2(n * G2W);
    memcpy(&g1[0], &A[h * G1W], h * G1W * 4);      // A2
    memcpy(&g2[0], &B[0], h * G2W * 4);            // B1
    memcpy(&g1[h * G1W], &A[0], h * G1W * 4);      // A1
    memcpy(&g2[h * G2W], &B[h * G2W], h * G2W * 4);  // B2
    uint32_t out[2 * 96];
    SIPP_TRY(sipp_inner_products(ctx, g1.data(), g2.data(), h, 2, out));
    memcpy(zl, out, 96 * 4);
    memcpy(zr, out + 96, 96 * 4);
    return SIPP_OK;
}

}  // namespace

extern "C" {

size_t sipp_native_proof_words(size_t n) {
    if (!is_pow2(n)) return 0;
    size_t lg = 0;
    while (((size_t)1 << lg) < n) lg++;
    return (2 * lg + 1) * F12W;
}

int sipp_prove_native(sipp_ctx* ctx, const uint32_t* A_in, const uint32_t* B_in, size_t n, uint32_t* proof) {
    if (!ctx || !A_in || !B_in || !proof) return SIPP_E_BADARG;
    if (!is_pow2(n)) return sipp_fail(ctx, SIPP_E_BADARG, "prove_native: n must be a power of two");
    std::vector<uint32_t> A(A_in, A_in + n * G1W), B(B_in, B_in + n * G2W);
    Transcript t;
    for (size_t i = 0; i < n; i++) {  // register A and B (prover_native.rs:35-39)
        t.append(&A[i * G1W], G1W);
        t.append(&B[i * G2W], G2W);
    }
    std::vector<uint32_t> msgs;  // in sending order; reversed at the end (prover_native.rs:78)
    uint32_t z[96], zl[96], zr[96];
    bool have_first = false, fold_started = false;
    std::vector<uint32_t> rec1, rec2;
    // the 255 doublings of A2_i / B2_i of a round do not depend on its challenge: they run on side streams while the pairing
    // products that determine the challenge are computed on the main stream
    auto begin_fold = [&]() -> int {
        fold_records(A, B, n, nullptr, nullptr, rec1, rec2);
        const int rc = sipp_fold_begin(ctx, rec1.data(), n / 2, rec2.data(), n / 2);
        fold_started = rc == SIPP_OK;
        return rc;
    };
    auto abandon_fold = [&]() {   // drain the side streams and hand the arena back on an error path
        if (fold_started) (void)sipp_fold_finish(ctx, rec1.data(), rec1.size() / SIPP_G1_IO_WORDS, rec2.data(), rec2.size() / SIPP_G2_IO_WORDS);
        fold_started = false;
    };
    if (n > 1) {
        SIPP_TRY(begin_fold());
        // Z = <A, B> and the first round's Z_L = <A2, B1>, Z_R = <A1, B2> depend on no challenge: ONE device pass
        const size_t h = n / 2;
        std::vector<uint32_t> g1(2 * n * G1W), g2(2 * n * G2W);
        memcpy(&g1[0], A.data(), n * G1W * 4);
        memcpy(&g2[0], B.data(), n * G2W * 4);
        memcpy(&g1[n * G1W], &A[h * G1W], h * G1W * 4);            // A2
        memcpy(&g2[n * G2W], &B[0], h * G2W * 4);                  // B1
        memcpy(&g1[(n + h) * G1W], &A[0], h * G1W * 4);            // A1
        memcpy(&g2[(n + h) * G2W], &B[h * G2W], h * G2W * 4);      // B2
        const uint32_t off[4] = {0, (uint32_t)n, (uint32_t)(n + h), (uint32_t)(2 * n)};
        uint32_t out[3 * 96];
        {
            const int rc = sipp_inner_products_groups(ctx, g1.data(), g2.data(), off, 3, out);
            if (rc != SIPP_OK) {
                abandon_fold();
                return rc;
            }
        }
        memcpy(z, out, 96 * 4);
        memcpy(zl, out + 96, 96 * 4);
        memcpy(zr, out + 192, 96 * 4);
        have_first = true;
    } else {
        SIPP_TRY(sipp_inner_product(ctx, A.data(), B.data(), n, z));
    }
    msgs.insert(msgs.end(), z, z + 96);
    t.append(z, 96);
    while (n > 1) {
        if (!fold_started) SIPP_TRY(begin_fold());
        if (!have_first) {
            const int rc = cross_products(ctx, A, B, n, zl, zr);
            if (rc != SIPP_OK) {
                abandon_fold();
                return rc;
            }
        }
        have_first = false;
        msgs.insert(msgs.end(), zl, zl + 96);
        t.append(zl, 96);
        msgs.insert(msgs.end(), zr, zr + 96);
        t.append(zr, 96);
        const U256 x = t.get_challenge();
        const U256 ix = inv_mod_r(x);
        uint32_t xw[8], ixw[8];
        to_u32(x, xw);
        to_u32(ix, ixw);
        fold_records(A, B, n, xw, ixw, rec1, rec2);
        fold_started = false;
        SIPP_TRY(sipp_fold_finish(ctx, rec1.data(), n / 2, rec2.data(), n / 2));
        take_outputs(rec1, rec2, n / 2, A, B);
        n /= 2;
    }
    const size_t cnt = msgs.size() / 96;
    for (size_t i = 0; i < cnt; i++) memcpy(proof + i * 96, &msgs[(cnt - 1 - i) * 96], 96 * 4);
    return SIPP_OK;
}

int sipp_verify_native(sipp_ctx* ctx, const uint32_t* A_in, const uint32_t* B_in, size_t n0, const uint32_t* proof,
                       uint32_t* statement, uint32_t* g1_ios, uint32_t* g2_ios, uint32_t* fq12_ios, int* accepted) {
    if (!ctx || !A_in || !B_in || !proof) return SIPP_E_BADARG;
    if (!is_pow2(n0)) return sipp_fail(ctx, SIPP_E_BADARG, "verify_native: n must be a power of two");
    size_t n = n0;
    std::vector<uint32_t> A(A_in, A_in + n * G1W), B(B_in, B_in + n * G2W);
    Transcript t;
    for (size_t i = 0; i < n; i++) {
        t.append(&A[i * G1W], G1W);
        t.append(&B[i * G2W], G2W);
    }
    size_t left = sipp_native_proof_words(n) / 96;  // the verifier pops from the END (verifier_native.rs:29, 38, 40)
    uint32_t Z[96], orig_Z[96];
    memcpy(orig_Z, proof + (--left) * 96, 96 * 4);
    memcpy(Z, orig_Z, 96 * 4);
    t.append(Z, 96);
    // every challenge follows from the messages alone: replay the transcript first ...
    size_t rounds = 0;
    while (((size_t)1 << rounds) < n) rounds++;
    std::vector<uint32_t> xs(rounds * 8), ixs(rounds * 8);
    std::vector<const uint32_t*> zls(rounds), zrs(rounds);
    for (size_t r = 0; r < rounds; r++) {
        zls[r] = proof + (--left) * 96;
        t.append(zls[r], 96);
        zrs[r] = proof + (--left) * 96;
        t.append(zrs[r], 96);
        const U256 x = t.get_challenge();
        to_u32(x, &xs[r * 8]);
        to_u32(inv_mod_r(x), &ixs[r * 8]);
    }
    // ... then all 2 log2 n powers Z_L^x, Z_R^(1/x) in ONE pass through the Fq12 chain kernel (offset 1), and the running
    // product Z <- Z Z_L^x Z_R^(1/x) on the host.  The obligations are {x: Z_L, offset: Z, exp_val: x} and
    // {x: Z_R, offset: Z Z_L^x, exp_val: 1/x} (verifier_circuit.rs:111-124).
    std::vector<uint32_t> f12rec(2 * rounds * SIPP_FQ12_IO_WORDS, 0);
    // the first round's doubling chains (side streams) run beside the Fq12 powers (main stream)
    std::vector<uint32_t> rec1_0, rec2_0;
    bool fold0 = false;
    if (rounds) {
        fold_records(A, B, n, &xs[0], &ixs[0], rec1_0, rec2_0);
        SIPP_TRY(sipp_fold_begin(ctx, rec1_0.data(), n / 2, rec2_0.data(), n / 2));
        fold0 = true;
        std::vector<uint32_t> pw(2 * rounds * SIPP_FQ12_IO_WORDS, 0);
        for (size_t r = 0; r < rounds; r++)
            for (int h = 0; h < 2; h++) {
                uint32_t* p = &pw[(2 * r + h) * SIPP_FQ12_IO_WORDS];
                memcpy(p, h ? zrs[r] : zls[r], 96 * 4);
                p[96] = 1;  // offset = 1
                memcpy(p + 192, h ? &ixs[r * 8] : &xs[r * 8], 32);
            }
        {
            const int rc = sipp_exp_outputs(ctx, SIPP_FQ12_EXP, pw.data(), 2 * rounds);
            if (rc != SIPP_OK) {
                (void)sipp_fold_finish(ctx, rec1_0.data(), n / 2, rec2_0.data(), n / 2);
                return rc;
            }
        }
        for (size_t r = 0; r < rounds; r++)
            for (int h = 0; h < 2; h++) {
                uint32_t* rec = &f12rec[(2 * r + h) * SIPP_FQ12_IO_WORDS];
                const uint32_t* p = &pw[(2 * r + h) * SIPP_FQ12_IO_WORDS];
                memcpy(rec, p, 96 * 4);
                memcpy(rec + 96, Z, 96 * 4);
                memcpy(rec + 192, p + 192, 32);
                f12_mul_host(Z, p + 200, rec + 200);
                memcpy(Z, rec + 200, 96 * 4);
            }
        if (fq12_ios) memcpy(fq12_ios, f12rec.data(), f12rec.size() * 4);
    }
    size_t o1 = 0, o2 = 0;
    for (size_t r = 0; r < rounds; r++) {
        std::vector<uint32_t> rec1, rec2;
        if (r == 0 && fold0) {
            rec1.swap(rec1_0);
            rec2.swap(rec2_0);
            SIPP_TRY(sipp_fold_finish(ctx, rec1.data(), n / 2, rec2.data(), n / 2));
        } else {
            SIPP_TRY(fold_round(ctx, A, B, n, &xs[r * 8], &ixs[r * 8], rec1, rec2));
        }
        if (g1_ios) memcpy(g1_ios + o1, rec1.data(), rec1.size() * 4);
        if (g2_ios) memcpy(g2_ios + o2, rec2.data(), rec2.size() * 4);
        o1 += rec1.size();
        o2 += rec2.size();
        take_outputs(rec1, rec2, n / 2, A, B);
        n /= 2;
    }
    if (statement) {  // A | B | Z | final_A | final_B | final_Z (statements.rs:24-39)
        uint32_t* s = statement;
        memcpy(s, A_in, n0 * G1W * 4);
        s += n0 * G1W;
        memcpy(s, B_in, n0 * G2W * 4);
        s += n0 * G2W;
        memcpy(s, orig_Z, 96 * 4);
        s += 96;
        memcpy(s, A.data(), G1W * 4);
        s += G1W;
        memcpy(s, B.data(), G2W * 4);
        s += G2W;
        memcpy(s, Z, 96 * 4);
    }
    if (accepted) {  // verifier_native.rs:80-84: pairing(final_A, final_B) == final_Z
        uint32_t e[96];
        SIPP_TRY(sipp_inner_product(ctx, A.data(), B.data(), 1, e));
        *accepted = memcmp(e, Z, sizeof e) == 0;
    }
    return SIPP_OK;
}

}  // extern "C"
