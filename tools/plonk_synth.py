#!/usr/bin/env python3
"""A synthetic plonky2-shaped circuit WITH gates, as data, and a witness for it -- the stand-in for the reference's outer circuit
(src/verifier_circuit.rs:213-226: CircuitConfig::standard_ecc_config, built by un-vendored crates) that tests and bench.py's
`outer_plonk` leg prove through sipp_plonk_prove_gates / orc_plonk_prove_gates.

Gate set (the styles plonky2's own gates come in; programs in the monomial format of include/sipp_hip.h, "gates as data"):
  0 Noop            no constraint
  1 Arithmetic      per op k: c0 w[4k] w[4k+1] + c1 w[4k+2] - w[4k+3]                      (ArithmeticGate, degree 3)
  2 BaseSum         sum_i 2^i limb_i - w[0];  limb_i^2 - limb_i, i < 32                    (BaseSumGate<2>, degree 2)
  3 PublicInput     w[i] - public_inputs_hash[i], i < 4                                    (PublicInputGate)
  4 SBox7           w[12+i] - (w[i] + c0)^7, i < 12                                        (the S-box of PoseidonGate, degree 7)
Selector group 0 (selector column 0) = gates 0 .. 3, group 1 (column 1) = gate 4: filters of degree 4 and 1, every filtered constraint
of degree <= 8 = max_degree.  Constant columns: 0, 1 selectors; 2, 3 the gates' constants c0, c1.
Everything vectorised over the rows with numpy (Goldilocks products from 32-bit halves): N = 2^18 x 136 wires takes seconds."""
import numpy as np

P = 0xFFFFFFFF00000001
M32 = np.uint64(0xFFFFFFFF)
EPS = np.uint64(0xFFFFFFFF)          # 2^64 mod p
UNUSED = 0xFFFFFFFF
PP = np.uint64(P)


def gl_add(a, b):
    s = a + b
    over = s < a
    s = np.where(over, s + EPS, s)                       # + 2^64 = + (2^32 - 1) mod p; cannot wrap again for canonical inputs
    return np.where(s >= PP, s - PP, s)


def gl_sub(a, b):
    d = a - b
    return np.where(a < b, d - EPS, d)                   # wrapped difference + p


def gl_mul(a, b):
    a = np.asarray(a, dtype=np.uint64)
    b = np.asarray(b, dtype=np.uint64)
    a0, a1, b0, b1 = a & M32, a >> np.uint64(32), b & M32, b >> np.uint64(32)
    ll, lh, hl, hh = a0 * b0, a0 * b1, a1 * b0, a1 * b1
    mid = lh + hl
    cmid = (mid < lh).astype(np.uint64)
    lo = ll + (mid << np.uint64(32))
    clo = (lo < ll).astype(np.uint64)
    hi = hh + (mid >> np.uint64(32)) + (cmid << np.uint64(32)) + clo
    h0, h1 = hi & M32, hi >> np.uint64(32)
    t0 = lo - h1                                          # lo - h1 (2^96 = -1)
    t0 = np.where(lo < h1, t0 - EPS, t0)
    t1 = (h0 << np.uint64(32)) - h0                       # h0 (2^32 - 1) < 2^64
    r = t0 + t1
    r = np.where(r < t1, r + EPS, r)
    return np.where(r >= PP, r - PP, r)


def gl_pow7(x):
    x2 = gl_mul(x, x)
    x4 = gl_mul(x2, x2)
    return gl_mul(gl_mul(x4, x2), x)


def rand_field(rng, shape):
    return (rng.integers(0, 1 << 63, size=shape, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=shape, dtype=np.uint64)) % PP


def powers(base, n):
    """[base^0 .. base^(n-1)], n a power of two, by doubling"""
    out = np.ones(n, dtype=np.uint64)
    m, b = 1, int(base)
    while m < n:
        out[m:2 * m] = gl_mul(out[:m], np.uint64(b))
        b = b * b % P
        m *= 2
    return out


def root_of_unity(log_n):
    return pow(1753635133440165772, 1 << (32 - log_n), P)


N_ARITH_OPS_MAX, N_LIMBS, N_SBOX = 20, 32, 12
GATE_NAMES = ["Noop", "Arithmetic", "BaseSum", "PublicInput", "SBox7"]


def circuit(num_wires=136, num_routed=80):
    """-> dict(num_wires, num_routed, num_constants, num_selectors, gates=[(selector_index, row, lo, hi, prog_offset, n_constraints)],
    programs=int64 array)"""
    assert num_routed >= 8 and num_wires >= max(num_routed, 2 * N_SBOX, N_LIMBS + 1)
    n_ops = min(N_ARITH_OPS_MAX, num_routed // 4)
    prog, gates = [], []

    def constraint(monos):
        prog.append(len(monos))
        for coef, factors in monos:
            prog.extend([coef, len(factors)])
            for kind, idx in factors:
                prog.extend([kind, idx])
    W, K, PIH = 0, 1, 2
    # 0 Noop
    gates.append((0, 0, 0, 4, len(prog), 0))
    # 1 Arithmetic
    off = len(prog)
    for k in range(n_ops):
        constraint([(1, [(K, 2), (W, 4 * k), (W, 4 * k + 1)]), (1, [(K, 3), (W, 4 * k + 2)]), (-1, [(W, 4 * k + 3)])])
    gates.append((0, 1, 0, 4, off, n_ops))
    # 2 BaseSum
    off = len(prog)
    constraint([(1 << i, [(W, 1 + i)]) for i in range(N_LIMBS)] + [(-1, [(W, 0)])])
    for i in range(N_LIMBS):
        constraint([(1, [(W, 1 + i), (W, 1 + i)]), (-1, [(W, 1 + i)])])
    gates.append((0, 2, 0, 4, off, 1 + N_LIMBS))
    # 3 PublicInput
    off = len(prog)
    for i in range(4):
        constraint([(1, [(W, i)]), (-1, [(PIH, i)])])
    gates.append((0, 3, 0, 4, off, 4))
    # 4 SBox7: out - sum_k C(7, k) in^k c0^(7 - k)
    off = len(prog)
    binom = [1, 7, 21, 35, 35, 21, 7, 1]
    for i in range(N_SBOX):
        constraint([(1, [(W, N_SBOX + i)])] + [(-binom[k], [(W, i)] * k + [(K, 2)] * (7 - k)) for k in range(8)])
    gates.append((1, 4, 4, 5, off, N_SBOX))
    return {"num_wires": num_wires, "num_routed": num_routed, "num_constants": 4, "num_selectors": 2, "n_arith_ops": n_ops, "gates": gates,
            "programs": np.array(prog, dtype=np.int64), "num_gate_constraints": max(g[5] for g in gates)}


def witness(circ, log_n, seed, pih):
    """wires [num_wires][N], constants_sigmas [4 + num_routed][N] (VALUES, natural row order) satisfying every gate and a random wire
    permutation over the gates' free input cells (cycles of three cells, constant on a cycle); pih = hash_no_pad(public inputs)"""
    rng = np.random.default_rng(seed)
    n = 1 << log_n
    Wn, R, n_ops = circ["num_wires"], circ["num_routed"], circ["n_arith_ops"]
    rows = np.arange(n)
    gate = np.array([1, 4, 1, 2, 4, 1, 0, 4, 1, 2], dtype=np.int64)[rows % 10]       # 40 % arithmetic, 30 % S-box, 20 % base sum, 10 % noop
    gate[0] = 3                                                                       # the public-input row
    wires = rand_field(rng, (Wn, n))
    c0, c1 = rand_field(rng, n), rand_field(rng, n)
    # free (routable) input cells: arithmetic inputs, S-box inputs, every routed wire of a noop row
    free = np.zeros((R, n), dtype=bool)
    ar, sb, no, bs = gate == 1, gate == 4, gate == 0, gate == 2
    for k in range(n_ops):
        free[4 * k:4 * k + 3, ar] = True
    free[:min(N_SBOX, R), sb] = True
    free[:, no] = True
    cells = np.flatnonzero(free.reshape(-1)).astype(np.int64)                          # index = column * n + row
    order = rng.permutation(cells)
    order = order[:len(order) - len(order) % 3].reshape(-1, 3)
    perm = np.arange(R * n, dtype=np.int64)
    perm[order[:, 0]], perm[order[:, 1]], perm[order[:, 2]] = order[:, 1], order[:, 2], order[:, 0]
    vals = rand_field(rng, order.shape[0])
    flat = wires[:R].reshape(-1)
    for q in range(3):
        flat[order[:, q]] = vals
    wires[:R] = flat.reshape(R, n)
    # outputs (on the rows of their gate only)
    ia, isb, ibs = np.flatnonzero(ar), np.flatnonzero(sb), np.flatnonzero(bs)
    ca0, ca1, cs0 = c0[ia], c1[ia], c0[isb]
    for k in range(n_ops):
        wires[4 * k + 3, ia] = gl_add(gl_mul(ca0, gl_mul(wires[4 * k, ia], wires[4 * k + 1, ia])), gl_mul(ca1, wires[4 * k + 2, ia]))
    for i in range(N_SBOX):
        wires[N_SBOX + i, isb] = gl_pow7(gl_add(wires[i, isb], cs0))
    bits = rng.integers(0, 2, size=(N_LIMBS, len(ibs)), dtype=np.uint64)
    total = np.zeros(len(ibs), dtype=np.uint64)
    for i in range(N_LIMBS):
        wires[1 + i, ibs] = bits[i]
        total = total + (bits[i] << np.uint64(i))
    wires[0, ibs] = total % PP
    for i in range(4):
        wires[i, 0] = np.uint64(int(pih[i]))
    # constants: selectors, gate constants; sigmas: k_col' w^row' of the cell a position maps to
    sel0 = np.where(gate < 4, gate, UNUSED).astype(np.uint64)
    sel1 = np.where(gate == 4, gate, UNUSED).astype(np.uint64)
    pw = powers(root_of_unity(log_n), n)
    ks = np.array([pow(7, j, P) for j in range(R)], dtype=np.uint64)
    sig = np.empty((R, n), dtype=np.uint64)
    pm = perm.reshape(R, n)
    for j in range(R):                                   # column by column: the temporaries of gl_mul stay in cache
        sig[j] = gl_mul(ks[pm[j] >> log_n], pw[pm[j] & (n - 1)])
    cs = np.concatenate([np.stack([sel0, sel1, c0, c1]), sig]).astype(np.uint64)
    return np.ascontiguousarray(wires), np.ascontiguousarray(cs), gate


def check_rows(circ, wires, cs, pih, rows):
    """plain-Python evaluation of every gate constraint on the given rows (selector value == gate index picks the gate): all zero?"""
    prog = [int(x) for x in circ["programs"]]
    for r in rows:
        for (si, row, lo, hi, off, nc) in circ["gates"]:
            if int(cs[si, r]) != row:
                continue
            w = off
            for _ in range(nc):
                nm = prog[w]
                w += 1
                s = 0
                for _m in range(nm):
                    t, nf = prog[w], prog[w + 1]
                    w += 2
                    for _f in range(nf):
                        kind, idx = prog[w], prog[w + 1]
                        w += 2
                        t = t * (int(wires[idx, r]) if kind == 0 else int(cs[idx, r]) if kind == 1 else int(pih[idx])) % P
                    s = (s + t) % P
                if s:
                    return False
    return True
