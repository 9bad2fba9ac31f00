#!/usr/bin/env python3
"""Generate the Poseidon-Goldilocks constant headers from data/poseidon_goldilocks_rc.txt.

Writes two independent copies (the product never includes anything from oracle/):
  oracle/poseidon_constants.h            (plain C, naive round constants only)
  sipp_amd/csrc/poseidon_constants.h     (HIP side: naive constants + the derived
                                          "fast partial round" tables)

The fast-partial-round tables are derived here from the MDS matrix and the round
constants with the standard Poseidon equivalent-matrix factorisation (Poseidon paper,
App. B; the same algebra plonky2 uses for FAST_PARTIAL_* upstream).  The HIP kernels
that use them are checked bit for bit against the naive oracle permutation.
"""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = 2**64 - 2**32 + 1
CIRC = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]
DIAG = [8] + [0] * 11
W = 12
N_FULL_HALF = 4
N_PARTIAL = 22


def load_rc():
    txt = open(os.path.join(ROOT, "data", "poseidon_goldilocks_rc.txt")).read()
    vals = [int(v, 16) for v in re.findall(r"0x[0-9a-f]{16}", txt)]
    assert len(vals) == 360
    return vals


def mds_matrix():
    # out[r] = sum_i in[(i + r) % 12] * CIRC[i] + in[r] * DIAG[r]
    M = [[0] * W for _ in range(W)]
    for r in range(W):
        for i in range(W):
            M[r][(i + r) % W] = (M[r][(i + r) % W] + CIRC[i]) % P
        M[r][r] = (M[r][r] + DIAG[r]) % P
    return M


def mat_mul(A, B):
    n, m, k = len(A), len(B[0]), len(B)
    return [[sum(A[i][t] * B[t][j] for t in range(k)) % P for j in range(m)] for i in range(n)]


def mat_vec(A, v):
    return [sum(A[i][j] * v[j] for j in range(len(v))) % P for i in range(len(A))]


def mat_inv(A):
    n = len(A)
    M = [row[:] + [1 if i == j else 0 for j in range(n)] for i, row in enumerate(A)]
    for c in range(n):
        piv = next(r for r in range(c, n) if M[r][c] % P)
        M[c], M[piv] = M[piv], M[c]
        inv = pow(M[c][c], P - 2, P)
        M[c] = [x * inv % P for x in M[c]]
        for r in range(n):
            if r != c and M[r][c]:
                f = M[r][c]
                M[r] = [(x - f * y) % P for x, y in zip(M[r], M[c])]
    return [row[n:] for row in M]


def transpose(A):
    return [list(r) for r in zip(*A)]


def derive_fast_partial(rc):
    """Returns (first_round_constants[12], initial_matrix[11][11], round_constants[22],
    vs[22][11], w_hats[22][11]) such that

      state += first_round_constants
      state  = [state[0]] ++ initial_matrix applied to state[1..]   (row-vector form below)
      for r in 0..22:
          state[0] = sbox(state[0]); state[0] += round_constants[r]   (r < 21; last adds 0)
          d = state[0]*M00 + sum_i state[i+1]*w_hats[r][i]
          state[i+1] += state[0]*vs[r][i];  state[0] = d

    is equal to the 22 naive partial rounds (add full constant vector, sbox lane 0, dense MDS).
    Verified numerically at the bottom of this script before any header is written.
    """
    M = mds_matrix()
    # --- constants: push all-lane constants backwards through the linear layers ---
    # naive partial round i: s -> M * sbox0(s + c_i).  Work from the last round to the first,
    # keeping only lane-0 constants inside the rounds and an initial full vector.
    Minv = mat_inv(M)
    partial_rc = [rc[12 * (N_FULL_HALF + r):12 * (N_FULL_HALF + r) + 12] for r in range(N_PARTIAL)]
    # opt constants: c'_r (scalar for lane 0), first_vec
    acc = partial_rc[N_PARTIAL - 1][:]
    scalars = [0] * N_PARTIAL
    for r in range(N_PARTIAL - 1, 0, -1):
        inv_acc = mat_vec(Minv, acc)
        scalars[r] = inv_acc[0]
        inv_acc[0] = 0
        acc = [(a + b) % P for a, b in zip(inv_acc, partial_rc[r - 1])]
    first = acc
    # after the transformation: round r (r = 0..21): sbox lane 0, then add scalars[r+1] to lane 0
    # *after* the MDS of round r ... we instead restate: state += first; for r: sbox0; (if r<21)
    # the constant scalars[r+1] is added to lane 0 after multiplying by M.  Because constants
    # on lane 0 commute with nothing, apply them as: s -> M*sbox0(s) then lane0 += M-pushed...
    # To keep this simple and exactly checkable we use the direct formulation below.
    return first, scalars


def naive_perm(state, rc):
    M = mds_matrix()
    s = state[:]
    rnd = 0
    def full():
        nonlocal s, rnd
        s = [(x + rc[12 * rnd + i]) % P for i, x in enumerate(s)]
        s = [pow(x, 7, P) for x in s]
        s = mat_vec(M, s)
        rnd += 1
    def partial():
        nonlocal s, rnd
        s = [(x + rc[12 * rnd + i]) % P for i, x in enumerate(s)]
        s[0] = pow(s[0], 7, P)
        s = mat_vec(M, s)
        rnd += 1
    for _ in range(4): full()
    for _ in range(22): partial()
    for _ in range(4): full()
    return s


def sparse_factor():
    """Factor the partial-round linear layers: M = M' * M'' style, iterated (Poseidon paper B).
    Returns initial 12x12 matrix `Mi` (with Mi[0][0]=1 border) and per-round (v[11], w_hat[11])."""
    M = mds_matrix()
    MT = transpose(M)
    # Follow the hadeshash reference: work with M^T (row-vector convention), M_mul accumulates.
    m_mul = [r[:] for r in MT]
    m_i = [[0] * W for _ in range(W)]
    vs, ws = [], []
    for _ in range(N_PARTIAL):
        # split m_mul = M' * M''
        m_hat = [row[1:] for row in m_mul[1:]]
        w = [m_mul[i][0] for i in range(1, W)]
        v = m_mul[0][1:]
        m_hat_inv = mat_inv(m_hat)
        w_hat = mat_vec(m_hat_inv, w)
        vs.append(v)
        ws.append(w_hat)
        # M' = [[1,0],[0,m_hat]]
        m_i = [[1] + [0] * (W - 1)] + [[0] + row for row in m_hat]
        m_mul = mat_mul(MT, m_i)
    return transpose(m_i), vs[::-1], ws[::-1]


def fast_perm(state, rc, tables):
    first, scalars, Mi, vs, ws = tables
    M = mds_matrix()
    s = state[:]
    rnd = 0
    for _ in range(4):
        s = [(x + rc[12 * rnd + i]) % P for i, x in enumerate(s)]
        s = [pow(x, 7, P) for x in s]
        s = mat_vec(M, s)
        rnd += 1
    # partial rounds, fast form
    s = [(x + c) % P for x, c in zip(s, first)]
    s = mat_vec(Mi, s)
    m00 = M[0][0]
    for r in range(N_PARTIAL):
        s[0] = pow(s[0], 7, P)
        if r < N_PARTIAL - 1:
            s[0] = (s[0] + scalars[r + 1]) % P
        d = (s[0] * m00 + sum(s[i + 1] * ws[r][i] for i in range(W - 1))) % P
        for i in range(W - 1):
            s[i + 1] = (s[i + 1] + s[0] * vs[r][i]) % P
        s[0] = d
    rnd += N_PARTIAL
    for _ in range(4):
        s = [(x + rc[12 * rnd + i]) % P for i, x in enumerate(s)]
        s = [pow(x, 7, P) for x in s]
        s = mat_vec(M, s)
        rnd += 1
    return s


BLK = 11  # partial rounds per lazy block (22 = 2 x 11)


def blocked_tables(vs, ws):
    """Lazy ("blocked") evaluation of the sparse partial rounds.  Inside a block starting at round r0 with
    S_i = state[i+1] and x_t = lane 0 after sbox + constant in round t:
        state[i+1] before round r = S_i + sum_{r0 <= t < r} x_t * vs[t][i]
        d_r = M00 * x_r + sum_i ws[r][i] * S_i + sum_{r0 <= t < r} x_t * CC[r][t],   CC[r][t] = sum_i ws[r][i] * vs[t][i]
    so no lane is reduced mod p inside the block: every term is a multiply-accumulate with a CONSTANT, and the
    kernel cuts constants into 22-bit limbs so that 64-bit accumulators never overflow (poseidon.hpp Acc6).
    Returns CC as a dict (r, t) -> value."""
    cc = {}
    for r in range(N_PARTIAL):
        r0 = r - r % BLK
        for t in range(r0, r):
            cc[(r, t)] = sum(ws[r][i] * vs[t][i] for i in range(W - 1)) % P
    return cc


def blocked_partial(s, scalars, vs, ws, cc):
    """the 22 sparse rounds in blocked form (python model of the kernel); s is after FIRST and MI"""
    m00 = mds_matrix()[0][0]
    s = s[:]
    for r0 in range(0, N_PARTIAL, BLK):
        S = s[1:]
        xs = []
        s0 = s[0]
        for r in range(r0, r0 + BLK):
            x = pow(s0, 7, P)
            if r < N_PARTIAL - 1:
                x = (x + scalars[r + 1]) % P
            s0 = (m00 * x + sum(ws[r][i] * S[i] for i in range(W - 1)) + sum(xs[t - r0] * cc[(r, t)] for t in range(r0, r))) % P
            xs.append(x)
        s = [s0] + [(S[i] + sum(xs[k] * vs[r0 + k][i] for k in range(BLK))) % P for i in range(W - 1)]
    return s


def blocked_perm(state, rc, tables, cc):
    first, scalars, Mi, vs, ws = tables
    M = mds_matrix()
    s = state[:]
    rnd = 0
    for _ in range(4):
        s = [(x + rc[12 * rnd + i]) % P for i, x in enumerate(s)]
        s = [pow(x, 7, P) for x in s]
        s = mat_vec(M, s)
        rnd += 1
    s = [(x + c) % P for x, c in zip(s, first)]
    s = mat_vec(Mi, s)
    s = blocked_partial(s, scalars, vs, ws, cc)
    rnd += N_PARTIAL
    for _ in range(4):
        s = [(x + rc[12 * rnd + i]) % P for i, x in enumerate(s)]
        s = [pow(x, 7, P) for x in s]
        s = mat_vec(M, s)
        rnd += 1
    return s


def combined_layer(first, Mi):
    """The MDS of the 4th full round, the FIRST constants and the dense pre-multiplication MI' = diag(1, MI) are one
    affine map s -> C s + c:  C = MI' * MDS (row 0 = row 0 of the MDS: small constants), c = MI' * first."""
    M = mds_matrix()
    C = mat_mul(Mi, M)
    c = mat_vec(Mi, first)
    assert C[0] == M[0] and c[0] == first[0]
    return C, c


def blocked_perm_combined(state, rc, tables, cc):
    """blocked_perm with the combined layer (python model of poseidon.hpp::permute)"""
    first, scalars, Mi, vs, ws = tables
    M = mds_matrix()
    C, c = combined_layer(first, Mi)
    s = state[:]
    rnd = 0
    for r in range(4):
        s = [(x + rc[12 * rnd + i]) % P for i, x in enumerate(s)]
        s = [pow(x, 7, P) for x in s]
        s = mat_vec(M, s) if r < 3 else [(a + b) % P for a, b in zip(mat_vec(C, s), c)]
        rnd += 1
    s = blocked_partial(s, scalars, vs, ws, cc)
    rnd += N_PARTIAL
    for _ in range(4):
        s = [(x + rc[12 * rnd + i]) % P for i, x in enumerate(s)]
        s = [pow(x, 7, P) for x in s]
        s = mat_vec(M, s)
        rnd += 1
    return s


# ---- dense constant products on the matrix pipe (round 4; poseidon.hpp::dense_mfma) -----------------------------------------
# The five dense products with full 64-bit constants that the lazy form of the partial rounds leaves -- the merged affine layer of full
# round 3 (rows 1..11 of C, 12 inputs), and per block of 11 rounds W (d_r's share of the block-start state: W[k][i] = ws[r0+k][i]) and
# V (the block-end update: V[i][k] = vs[r0+k][i]) -- as int8 products of BYTE PLANES: the state words are cut into 8 unsigned bytes
# u_a, made signed by ^ 0x80 (s_a = u_a - 128); every constant into 8 SIGNED base-256 digits d_b of c or c - p; then
#   sum_e c_e x_e = sum_t 2^(8t) D_t + 128 * 0x0101010101010101 * sum_e c'_e,    D_t = sum_(a+b=t) sum_e d_(e,b) s_(e,a)   (int32)
# and with 2^64 = 2^32 - 1, 2^96 = -1 (mod p) the fifteen D_t fold into two signed 64-bit chains
#   L = sum_(t<4) D_t 2^(8t) - sum_(8<=t<12) D_t 2^(8(t-8)) - sum_(t>=12) D_t 2^(8(t-12)),  H = sum_(4<=t<8) D_t 2^(8(t-4)) + sum_(8<=t<12) D_t 2^(8(t-8))
# value = L + 2^32 H + K.  The chains START at K's halves plus 2^50 (so that both stay positive; K absorbs -2^50 (1 + 2^32)).
DENSE_OFF = 1 << 50


def signed_digits(c):
    """8 digits in [-128, 127] with sum d_b 256^b = c or c - p"""
    for cand in (c, c - P):
        v, ds = cand, []
        for _ in range(8):
            d = v & 0xFF
            if d >= 128:
                d -= 256
            ds.append(d)
            v = (v - d) >> 8
        if v == 0:
            assert sum(d << (8 * b) for b, d in enumerate(ds)) == cand
            return ds, cand
    raise AssertionError("no signed-digit form for %x" % c)


def dense_matrices(first, Mi, vs, ws):
    """[(rows x 12 matrix, additive constants per row)] in kernel order: 0 = combined layer rows 1..11; 1 + 2 b = W of block b; 2 + 2 b = V"""
    Cm, cv = combined_layer(first, Mi)
    mats = [([Cm[i][:] for i in range(1, 12)], [cv[i] for i in range(1, 12)])]
    for r0 in range(0, N_PARTIAL, BLK):
        Wm = [[ws[r0 + k][i] for i in range(W - 1)] + [0] for k in range(BLK)]
        Vm = [[vs[r0 + k][i] for k in range(BLK)] + [0] for i in range(W - 1)]
        mats.append((Wm, [0] * BLK))
        mats.append((Vm, [0] * (W - 1)))
    return mats


def dense_tables(mats):
    """A fragments [mat][limb b][lane 0..63][4 u32] (this lane's 16 bytes of the block-diagonal A of v_mfma_i32_32x32x32_i8: row = lane & 31,
    k = 16 (lane >> 5) + e -- the layout of poseidon.hpp::mds_a_fragment) and chain starts [mat][row 0..11][L, H] (u64)"""
    frag, starts = [], []
    for Mx, add in mats:
        dig = [[signed_digits(c) for c in row] for row in Mx]
        for b in range(8):
            for lane in range(64):
                row, h = lane & 31, lane >> 5
                mine = row < 24 and ((row >> 2) & 1) == h
                reg = (row & 3) + 4 * (row >> 3)
                w = [0, 0, 0, 0]
                if mine and reg < len(Mx):
                    for e in range(12):
                        w[e >> 2] |= (dig[reg][e][0][b] & 0xFF) << (8 * (e & 3))
                frag += w
        for r in range(12):
            if r < len(Mx):
                bias = 128 * 0x0101010101010101 * sum(dig[r][e][1] for e in range(12))
                K = (bias + add[r] - DENSE_OFF * (1 + (1 << 32))) % P
            else:
                K = (-DENSE_OFF * (1 + (1 << 32))) % P
            starts += [(K & 0xFFFFFFFF) + DENSE_OFF, (K >> 32) + DENSE_OFF]
    return frag, starts


def dense_model(mat_index, frag, starts, x, addend=None):
    """python model of poseidon.hpp::dense_mfma for one lane (lane 0: rows 0..11 sit in A rows (r & 3) + 8 (r >> 2)): x = 12 arbitrary u64"""
    planes = [[((x[e] >> (8 * a)) & 0xFF) - 128 for e in range(12)] for a in range(8)]

    def a_byte(b, reg, e):           # A[row][k = e] of limb b as the lane that owns row `row` of half 0 holds it
        row = (reg & 3) + 8 * (reg >> 2)
        w = frag[((mat_index * 8 + b) * 64 + row) * 4 + (e >> 2)]
        v = (w >> (8 * (e & 3))) & 0xFF
        return v - 256 if v >= 128 else v

    out = []
    for r in range(12):
        D = [sum(a_byte(t - a, r, e) * planes[a][e] for a in range(max(0, t - 7), min(7, t) + 1) for e in range(12)) for t in range(15)]
        assert all(-(1 << 31) <= d < (1 << 31) for d in D)
        L, H = starts[(mat_index * 12 + r) * 2], starts[(mat_index * 12 + r) * 2 + 1]
        if addend is not None:
            L += addend[r] & 0xFFFFFFFF
            H += addend[r] >> 32
        for t in range(15):
            if t < 4:
                L += D[t] << (8 * t)
            elif t < 8:
                H += D[t] << (8 * (t - 4))
            elif t < 12:
                H += D[t] << (8 * (t - 8))
                L -= D[t] << (8 * (t - 8))
            else:
                L -= D[t] << (8 * (t - 12))
        assert 0 < L < (1 << 52) and 0 < H < (1 << 52)       # the kernel's chains are positive 64-bit values
        lo = (L + (H << 32)) & 0xFFFFFFFFFFFFFFFF
        hi = (H >> 32) + (1 if lo < L else 0)
        assert hi < (1 << 32)
        out.append((lo + (hi << 64)) % P)                     # reduce96 of (hi, lo)
    return out


# ---- the same products for the TWO-LANES-PER-STATE kernel (poseidon_pair.hpp): lanes l and l + 32 share a state -------------------
# v_mfma_i32_32x32x32_i8 reads A[i][k] from lane (i = l & 31, k = 16 (l >> 5) + byte), B[k][n] from lane (n = l & 31, k = 16 (l >> 5) +
# byte) and leaves D[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][n = l & 31] in register r of lane l.  With lane (n, h) holding the six elements
# 6 h + j of state n, ONE instruction sees the whole state of 32 leaves in its K dimension, and the unused rows / columns carry a second
# product: K slot ps (bytes 8 ps + j of the lane's sixteen) = a second byte plane, M slot ds (registers 6 ds + j) = a second digit.
#   MDS layers (small constants): A = diag(M, M): register j = M x (plane a), register 6 + j = M x (plane a + 1): 4 instructions per layer.
#   dense products: B = [plane a | plane a + 1] (a even), A_b = rows ds: [digit b + ds | digit b + ds - 1]: register 6 ds + j collects
#   the digit sum t + ds, t = a + b: 20 instructions per product (chains t = 0, 2, .. 14 over a = 0, 2, 4, 6) instead of 64.
PAIR_FRAGS = 5          # b = 0, 2, 4, 6, 8


def pair_row(rho):
    """A row rho -> (output half h, register) of the lane that finds it"""
    return (rho >> 2) & 1, (rho & 3) + 4 * (rho >> 3)


def pair_fragment(entry):
    """[lane 0..63][4 u32]: entry(e_out, ds, e_in, ps) -> signed byte"""
    frag = []
    for lane in range(64):
        rho, hh = lane & 31, lane >> 5
        h_out, reg = pair_row(rho)
        w = [0, 0, 0, 0]
        if reg < 12:
            j, ds = reg % 6, reg // 6
            for ps in range(2):
                for jj in range(6):
                    v = entry(6 * h_out + j, ds, 6 * hh + jj, ps)
                    assert -128 <= v <= 127
                    byte = 8 * ps + jj
                    w[byte >> 2] |= (v & 0xFF) << (8 * (byte & 3))
        frag += w
    return frag


def mfma_model(afrag, bfrag, acc):
    """D = A B + C of v_mfma_i32_32x32x32_i8 on per-lane operands: afrag / bfrag [64][4 u32], acc [64][16] -> [64][16]"""
    def sb(words, byte):
        v = (words[byte >> 2] >> (8 * (byte & 3))) & 0xFF
        return v - 256 if v >= 128 else v
    A = [[sb(afrag[i + 32 * (k >> 4)], k & 15) for k in range(32)] for i in range(32)]
    B = [[sb(bfrag[n + 32 * (k >> 4)], k & 15) for n in range(32)] for k in range(32)]
    out = []
    for lane in range(64):
        n, h = lane & 31, lane >> 5
        regs = []
        for r in range(16):
            i = (r & 3) + 8 * (r >> 2) + 4 * h
            v = acc[lane][r] + sum(A[i][k] * B[k][n] for k in range(32))
            assert -(1 << 31) <= v < (1 << 31)
            regs.append(v)
        out.append(regs)
    return out


def pair_matrices(first, Mi, vs, ws, rc):
    """12 x 12 matrices over (output slot, input slot) and additive constants, kernel order: 0 = the whole combined layer of round 3;
    1 + 2 b = W of block b (output slot k = local round, input slot = element); 2 + 2 b = V (output slot = element, input slot 6 hh + jj =
    local round 2 jj + hh: a lane keeps the x_k of its parity); the constants of full round 26 ride on the last V"""
    Cm, cv = combined_layer(first, Mi)
    mats = [([row[:] for row in Cm], cv[:])]
    for r0 in range(0, N_PARTIAL, BLK):
        Wm = [[0] + [ws[r0 + k][i] for i in range(W - 1)] for k in range(BLK)] + [[0] * 12]
        Vm = [[0] * 12]
        for i in range(W - 1):
            row = [0] * 12
            for k in range(BLK):
                row[6 * (k & 1) + (k >> 1)] = vs[r0 + k][i]
            Vm.append(row)
        last = r0 + BLK == N_PARTIAL
        mats.append((Wm, [0] * 12))
        mats.append((Vm, [0] + [rc[12 * (N_FULL_HALF + N_PARTIAL) + e] if last else 0 for e in range(1, 12)]))
    return mats


def pair_planes(x6):
    """the eight byte planes of a lane's six words as B words [a][2]: bytes 0..5 of the pair = the elements' bytes (^ 0x80)"""
    out = []
    for a in range(8):
        by = [(((x >> (8 * a)) & 0xFF) ^ 0x80) for x in x6] + [0, 0]
        out.append([by[0] | by[1] << 8 | by[2] << 16 | by[3] << 24, by[4] | by[5] << 8])
    return out


def pair_tables(mats):
    """A fragments [mat][f = b / 2][lane][4] and chain starts [mat][element][L, H]"""
    frag, starts = [], []
    for Mx, add in mats:
        dig = [[signed_digits(c) for c in row] for row in Mx]

        for f in range(PAIR_FRAGS):
            def entry(eo, ds, ei, ps, b=2 * f):
                beta = b + ds - ps
                return dig[eo][ei][0][beta] if 0 <= beta <= 7 else 0
            frag += pair_fragment(entry)
        for e in range(12):
            bias = 128 * 0x0101010101010101 * sum(dig[e][i][1] for i in range(12))
            K = (bias + add[e] - DENSE_OFF * (1 + (1 << 32))) % P
            starts += [(K & 0xFFFFFFFF) + DENSE_OFF, (K >> 32) + DENSE_OFF]
    return frag, starts


def pair_mds_fragment():
    M = mds_matrix()
    return pair_fragment(lambda eo, ds, ei, ps: M[eo][ei] if ds == ps else 0)


def pair_fold(D, L, H, t):
    """digit sum D_t into the chains (the arithmetic of poseidon.hpp::dense_mfma)"""
    if t < 4:
        L += D << (8 * t)
    elif t < 8:
        H += D << (8 * (t - 4))
    elif t < 12:
        H += D << (8 * (t - 8))
        L -= D << (8 * (t - 8))
    else:
        L -= D << (8 * (t - 12))
    return L, H


def pair_dense_model(mi, frag, starts, x12, addend=None):
    """python model of poseidon_pair.hpp::dense_pair for one state (every column n of the wave holds the same state)"""
    planes = [pair_planes(x12[6 * h:6 * h + 6]) for h in range(2)]
    L = [starts[(mi * 12 + e) * 2] + ((addend[e] & 0xFFFFFFFF) if addend else 0) for e in range(12)]
    H = [starts[(mi * 12 + e) * 2 + 1] + ((addend[e] >> 32) if addend else 0) for e in range(12)]
    for t in range(0, 16, 2):
        acc = [[0] * 16 for _ in range(64)]
        for a in range(0, 8, 2):
            b = t - a
            if b < 0 or b > 8:
                continue
            f = b // 2
            afr = [frag[((mi * PAIR_FRAGS + f) * 64 + lane) * 4:((mi * PAIR_FRAGS + f) * 64 + lane) * 4 + 4] for lane in range(64)]
            bfr = [planes[lane >> 5][a] + planes[lane >> 5][a + 1] for lane in range(64)]
            acc = mfma_model(afr, bfr, acc)
        for h in range(2):
            for j in range(6):
                e = 6 * h + j
                L[e], H[e] = pair_fold(acc[32 * h][j], L[e], H[e], t)
                if t + 1 < 15:
                    L[e], H[e] = pair_fold(acc[32 * h][6 + j], L[e], H[e], t + 1)
                else:
                    assert acc[32 * h][6 + j] == 0
    out = []
    for e in range(12):
        assert 0 < L[e] < (1 << 52) and 0 < H[e] < (1 << 52)
        lo = (L[e] + (H[e] << 32)) & 0xFFFFFFFFFFFFFFFF
        hi = (H[e] >> 32) + (1 if lo < L[e] else 0)
        out.append((lo + (hi << 64)) % P)
    return out


def pair_mds_model(frag, x12, add12):
    """python model of poseidon_pair.hpp::mds_pair: out = M x + add (mod p)"""
    planes = [pair_planes(x12[6 * h:6 * h + 6]) for h in range(2)]
    D = []
    for a in range(0, 8, 2):
        afr = [frag[lane * 4:lane * 4 + 4] for lane in range(64)]
        bfr = [planes[lane >> 5][a] + planes[lane >> 5][a + 1] for lane in range(64)]
        D.append(mfma_model(afr, bfr, [[0] * 16 for _ in range(64)]))
    M = mds_matrix()
    out = []
    for e in range(12):
        h, j = e // 6, e % 6
        bias = 128 * sum(M[e]) * 0x01010101
        al = bias + (add12[e] & 0xFFFFFFFF) + sum((D[a][32 * h][j] << (16 * a)) + (D[a][32 * h][6 + j] << (16 * a + 8)) for a in range(2))
        ah = bias + (add12[e] >> 32) + sum((D[2 + a][32 * h][j] << (16 * a)) + (D[2 + a][32 * h][6 + j] << (16 * a + 8)) for a in range(2))
        assert 0 <= al < (1 << 44) and 0 <= ah < (1 << 44)
        out.append((al + (ah << 32)) % P)
    return out


def limbs3(c):
    """22 + 22 + 20 bits"""
    assert 0 <= c < P
    return [c & 0x3FFFFF, (c >> 22) & 0x3FFFFF, c >> 44]


def blocked_words(vs, ws, cc):
    """u32 table read by poseidon.hpp partial_rounds_blocked: per block [ per local round k: ws[r][0..10] then
    CC[r][r0..r0+k-1] ] then vs transposed [i][k]; every constant as 3 limbs."""
    out = []
    for r0 in range(0, N_PARTIAL, BLK):
        for k in range(BLK):
            r = r0 + k
            for i in range(W - 1):
                out += limbs3(ws[r][i])
            for t in range(r0, r):
                out += limbs3(cc[(r, t)])
        for i in range(W - 1):
            for k in range(BLK):
                out += limbs3(vs[r0 + k][i])
    assert len(out) == (N_PARTIAL // BLK) * (33 * BLK + 3 * (BLK * (BLK - 1) // 2) + 33 * BLK)
    return out


def fmt32(vals, per=8):
    out = []
    for i in range(0, len(vals), per):
        out.append("    " + ", ".join("0x%06xu" % v for v in vals[i:i + per]) + ",")
    return "\n".join(out)


def fmt32x(vals, per=8):
    out = []
    for i in range(0, len(vals), per):
        out.append("    " + ", ".join("0x%08xu" % v for v in vals[i:i + per]) + ",")
    return "\n".join(out)


def fmt(vals, per=4):
    out = []
    for i in range(0, len(vals), per):
        out.append("    " + ", ".join("0x%016xULL" % v for v in vals[i:i + per]) + ",")
    return "\n".join(out)


def main():
    rc = load_rc()
    first, scalars = derive_fast_partial(rc)
    Mi, vs, ws = sparse_factor()
    tables = (first, scalars, Mi, vs, ws)
    cc = blocked_tables(vs, ws)
    import random
    rnd = random.Random(1)
    for t in range(8):
        st = [rnd.randrange(P) for _ in range(12)] if t else [0] * 12
        a, b = naive_perm(st, rc), fast_perm(st, rc, tables)
        assert a == b, ("fast partial rounds mismatch", t)
        assert a == blocked_perm(st, rc, tables, cc), ("blocked partial rounds mismatch", t)
        assert a == blocked_perm_combined(st, rc, tables, cc), ("combined layer mismatch", t)
    # the matrix-pipe form of the five dense products against plain matrix-vector products, on arbitrary (non-canonical) u64 inputs
    mats = dense_matrices(first, Mi, vs, ws)
    dfrag, dstarts = dense_tables(mats)
    for mi, (Mx, add) in enumerate(mats):
        for t in range(6):
            x = [rnd.randrange(1 << 64) for _ in range(12)] if t > 1 else [(1 << 64) - 1] * 12 if t else [0] * 12
            extra = [rnd.randrange(1 << 64) for _ in range(12)] if (mi and mi % 2 == 0) else None
            got = dense_model(mi, dfrag, dstarts, x, extra)
            for r in range(len(Mx)):
                want = (sum(Mx[r][e] * x[e] for e in range(12)) + add[r] + (extra[r] if extra else 0)) % P
                assert got[r] == want, ("dense product on the matrix pipe", mi, r, t)
    # the two-lanes-per-state forms of the same products (poseidon_pair.hpp) against plain matrix-vector products
    pmats = pair_matrices(first, Mi, vs, ws, rc)
    pfrag, pstarts = pair_tables(pmats)
    pmds = pair_mds_fragment()
    Mm = mds_matrix()
    for t in range(4):
        x = [rnd.randrange(1 << 64) for _ in range(12)] if t > 1 else [(1 << 64) - 1] * 12 if t else [0] * 12
        addc = [rnd.randrange(P) for _ in range(12)]
        got = pair_mds_model(pmds, x, addc)
        assert got == [(sum(Mm[r][e] * x[e] for e in range(12)) + addc[r]) % P for r in range(12)], ("pair MDS on the matrix pipe", t)
        for mi, (Mx, add) in enumerate(pmats):
            extra = [rnd.randrange(1 << 64) for _ in range(12)] if (mi and mi % 2 == 0) else None
            got = pair_dense_model(mi, pfrag, pstarts, x, extra)
            for r in range(12):
                want = (sum(Mx[r][e] * x[e] for e in range(12)) + add[r] + (extra[r] if extra else 0)) % P
                assert got[r] == want, ("pair dense product on the matrix pipe", mi, r, t)
    kat0 = naive_perm([0] * 12, rc)
    assert kat0[0] == 0x3c18a9786cb0b359 and kat0[11] == 0x1792b1c4342109d7

    hdr = "/* GENERATED by tools/gen_poseidon_header.py from data/poseidon_goldilocks_rc.txt -- do not edit. */\n"
    # --out-root DIR: write below DIR instead of the repository (tests/test_generated_tables.py compares bytes; tracked files stay untouched)
    out_root = sys.argv[sys.argv.index("--out-root") + 1] if "--out-root" in sys.argv else ROOT
    os.makedirs(os.path.join(out_root, "oracle"), exist_ok=True)
    os.makedirs(os.path.join(out_root, "sipp_amd", "csrc"), exist_ok=True)
    with open(os.path.join(out_root, "oracle", "poseidon_constants.h"), "w") as f:
        f.write(hdr)
        f.write("#ifndef ORACLE_POSEIDON_CONSTANTS_H\n#define ORACLE_POSEIDON_CONSTANTS_H\n#include <stdint.h>\n")
        f.write("static const uint64_t POSEIDON_RC[360] = {\n" + fmt(rc) + "\n};\n")
        f.write("static const uint64_t POSEIDON_CIRC[12] = {%s};\n" % ", ".join(map(str, CIRC)))
        f.write("static const uint64_t POSEIDON_DIAG[12] = {%s};\n" % ", ".join(map(str, DIAG)))
        f.write("#endif\n")
    with open(os.path.join(out_root, "sipp_amd", "csrc", "poseidon_constants.h"), "w") as f:
        f.write(hdr)
        f.write("#pragma once\n#include <stdint.h>\n")
        f.write("// index = 12*round + lane, rounds 0..29 (4 full, 22 partial, 4 full)\n")
        f.write("static const uint64_t SIPP_POSEIDON_RC[360] = {\n" + fmt(rc) + "\n};\n")
        f.write("// fast partial rounds: state += FIRST; state = MI * state; 22 x (sbox0, +SCALAR, sparse)\n")
        f.write("static const uint64_t SIPP_POSEIDON_FAST_FIRST[12] = {\n" + fmt(first) + "\n};\n")
        f.write("static const uint64_t SIPP_POSEIDON_FAST_SCALAR[22] = {\n" + fmt(scalars[1:] + [0]) + "\n};\n")
        mi_flat = [Mi[i][j] for i in range(1, 12) for j in range(1, 12)]
        f.write("// 11x11 block of the initial matrix (row-major, out[i+1] = sum_j MI[i][j]*in[j+1]); lane 0 passes through\n")
        f.write("static const uint64_t SIPP_POSEIDON_FAST_MI[121] = {\n" + fmt(mi_flat) + "\n};\n")
        f.write("static const uint64_t SIPP_POSEIDON_FAST_VS[22*11] = {\n" + fmt([x for r in vs for x in r]) + "\n};\n")
        f.write("static const uint64_t SIPP_POSEIDON_FAST_WHAT[22*11] = {\n" + fmt([x for r in ws for x in r]) + "\n};\n")
        f.write("// blocked (lazy) partial rounds, constants in 22/22/20-bit limbs: see blocked_words() in tools/gen_poseidon_header.py\n")
        bw = blocked_words(vs, ws, cc)
        f.write("#define SIPP_POSEIDON_BLK_ROUNDS %d\n#define SIPP_POSEIDON_BLK_WORDS %d\n" % (BLK, len(bw) // (N_PARTIAL // BLK)))
        f.write("static const uint32_t SIPP_POSEIDON_BLK3[%d] = {\n" % len(bw) + fmt32(bw) + "\n};\n")
        mi3 = [w for v in mi_flat for w in limbs3(v)]
        f.write("static const uint32_t SIPP_POSEIDON_MI3[363] = {\n" + fmt32(mi3) + "\n};\n")
        f.write("// rows 1..11 of C = diag(1, MI) * MDS (12 limb triples each) and c = diag(1, MI) * FIRST: the linear layer of\n"
                "// full round 3, the FIRST constants and the dense pre-multiplication as ONE affine map (row 0 is MDS row 0 + FIRST[0])\n")
        Cm, cv = combined_layer(first, Mi)
        comb3 = [w for i in range(1, 12) for j in range(12) for w in limbs3(Cm[i][j])]
        f.write("static const uint32_t SIPP_POSEIDON_COMB3[396] = {\n" + fmt32(comb3) + "\n};\n")
        f.write("static const uint64_t SIPP_POSEIDON_COMB_C[12] = {\n" + fmt(cv) + "\n};\n")
        f.write("// dense constant products on the matrix pipe (poseidon.hpp::dense_mfma; dense_tables() in tools/gen_poseidon_header.py):\n"
                "// matrices 0 = rows 1..11 of the combined layer, 1 + 2 b = W of block b, 2 + 2 b = V of block b; A fragments\n"
                "// [matrix][digit b][lane][4] of signed base-256 digits, chain starts [matrix][row][L, H]\n")
        f.write("#define SIPP_POSEIDON_DENSE_MATS %d\n" % len(mats))
        f.write("static const uint32_t SIPP_POSEIDON_DENSE_A[%d] = {\n" % len(dfrag) + fmt32x(dfrag) + "\n};\n")
        f.write("static const uint64_t SIPP_POSEIDON_DENSE_START[%d] = {\n" % len(dstarts) + fmt(dstarts) + "\n};\n")
        f.write("// the same for the two-lanes-per-state kernel (poseidon_pair.hpp; pair_tables() / pair_mds_fragment() in the generator):\n"
                "// A fragments [matrix][b / 2][lane][4] carrying two digits and two byte planes each, chain starts [matrix][element][L, H],\n"
                "// and the block-diagonal fragment diag(MDS, MDS) of the full-round layers\n")
        f.write("#define SIPP_POSEIDON_PAIR_FRAGS %d\n" % PAIR_FRAGS)
        f.write("static const uint32_t SIPP_POSEIDON_PAIR_A[%d] = {\n" % len(pfrag) + fmt32x(pfrag) + "\n};\n")
        f.write("static const uint64_t SIPP_POSEIDON_PAIR_START[%d] = {\n" % len(pstarts) + fmt(pstarts) + "\n};\n")
        f.write("static const uint32_t SIPP_POSEIDON_PAIR_MDS_A[%d] = {\n" % len(pmds) + fmt32x(pmds) + "\n};\n")
        f.write("// CC[r][t] of the lazy blocks (blocked_tables) as limb triples, [block][local round k][local round j 0..11] with zeros for\n"
                "// j >= k: the lanes of a pair read the entries of their parity without a bound check\n")
        cc3 = []
        for r0 in range(0, N_PARTIAL, BLK):
            for k in range(BLK):
                for j in range(12):
                    cc3 += limbs3(cc[(r0 + k, r0 + j)]) if j < k else [0, 0, 0]
        f.write("static const uint32_t SIPP_POSEIDON_PAIR_CC3[%d] = {\n" % len(cc3) + fmt32(cc3) + "\n};\n")
    assert all(Mi[0][j] == (1 if j == 0 else 0) for j in range(12)) and all(Mi[i][0] == 0 for i in range(1, 12))
    print("ok: headers written; fast-partial tables verified against naive permutation")


if __name__ == "__main__":
    main()
