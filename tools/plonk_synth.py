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

Round 6: with the column counts of CircuitConfig::standard_ecc_config (>= 130 wires, >= 80 routed) the circuit is
RECURSION-SHAPED instead -- the gate kinds a plonky2 recursive verifier is made of, 118 gate constraints per row slot:
  group 0 (selector column 0): 0 Noop, 1 Arithmetic, 2 BaseSum, 3 PublicInput,
                               4 Constant      w[i] - c_i, i < 2                                                     (ConstantGate)
  group 1 (selector column 1): 5 U32MulAdd     per op: a b + c - lo - 2^32 hi;  lo / hi = sum 4^i limb_i;  prod_{j<4}(limb - j)   (plonky2_u32's
                                                U32ArithmeticGate: 2-bit limbs, degree 4; 3 ops, 105 constraints)
                               6 RandomAccess  per copy: bits boolean, index = b0 + 2 b1, claimed = list[index] by two folds      (RandomAccessGate,
                                                bits = 2: degree 3; 10 copies, 40 constraints)
                               7 Reducing      acc_i = acc_{i-1} alpha + c_i over the quadratic extension, 40 base coefficients    (ReducingGate: 80)
  group 2 (selector column 2): 8 Poseidon      the WHOLE permutation of a 12-wide state: in, out, and one wire per S-box input of rounds 1 .. 29
                                                (round 0 reads in + rc), every constraint of degree 7 over the wires                (PoseidonGate: 118;
                                                the real round constants and MDS matrix: oracle/py/plonky2_generic.py)
Constant columns: 0 .. 2 selectors; 3, 4 the gates' constants.  Filters of degree 5 / 3 / 1: every filtered constraint of degree <= 8.
Everything vectorised over the rows with numpy (Goldilocks products from 32-bit halves): N = 2^18 x 136 wires takes about ten seconds
(the Poseidon rows: 30 rounds of a 12 x 12 matrix)."""
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
    if num_wires >= 130 and num_routed >= 80:
        return circuit_recursion_shaped(num_wires, num_routed)
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
            "programs": np.array(prog, dtype=np.int64), "num_gate_constraints": max(g[5] for g in gates), "gate_names": GATE_NAMES, "rich": False}


# ------------------------------------------------------------------------------------------------------------------------------------
# the recursion-shaped gate set (round 6)
RICH_GATE_NAMES = ["Noop", "Arithmetic", "BaseSum", "PublicInput", "Constant", "U32MulAdd", "RandomAccess", "Reducing", "Poseidon"]
U32_OPS, U32_LIMBS, U32_STRIDE = 3, 16, 5 + 2 * 16          # per op: a, b, c, lo, hi, 16 + 16 two-bit limbs
RA_COPIES, RA_STRIDE = 10, 8                                # per copy: index, claimed, x0 .. x3, b0, b1
RED_K = 40                                                  # Reducing: alpha (2), old acc (2), RED_K coefficients, RED_K accumulators (2 each)
POS_IN, POS_OUT, POS_SBOX = 0, 12, 24                       # Poseidon: in[12], out[12], S-box input wires of rounds 1 .. 29 (36 + 22 + 48)
EXT_W = 7                                                   # F[X] / (X^2 - 7)


def _i64(c):
    """a field element as the int64 coefficient of a program word (negative = p - |c|)"""
    c %= P
    return c if c < (1 << 63) else c - P


_POSEIDON_TABLES = None


def _poseidon_tables():
    """(ALL_ROUND_CONSTANTS[360], MDS rows, None): the constants from data/poseidon_goldilocks_rc.txt (the file the product's and the oracle's
    headers are generated from) and the circulant-plus-diagonal MDS matrix -- read here directly, so that building the stand-in circuit
    (bench.py's outer_plonk leg) imports nothing from oracle/"""
    global _POSEIDON_TABLES
    if _POSEIDON_TABLES is None:
        import os
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        vals = []
        for line in open(os.path.join(root, "data", "poseidon_goldilocks_rc.txt")):
            vals += [int(t, 16) for t in line.split("#")[0].replace(",", " ").split()]
        assert len(vals) == 360
        circ, diag = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20], [8] + [0] * 11
        rows = [tuple(circ[(c - r) % 12] + (diag[r] if c == r else 0) for c in range(12)) for r in range(12)]
        _POSEIDON_TABLES = (vals, rows, None)
    return _POSEIDON_TABLES


def poseidon_sbox_wire(rnd, i):
    """wire that holds the S-box INPUT of state element i in round rnd (1 .. 29); round 0 has none (its inputs are in + rc)"""
    if 1 <= rnd <= 3:
        return POS_SBOX + 12 * (rnd - 1) + i
    if 4 <= rnd <= 25:
        assert i == 0
        return POS_SBOX + 36 + (rnd - 4)
    assert 26 <= rnd <= 29
    return POS_SBOX + 36 + 22 + 12 * (rnd - 26) + i


def _poseidon_constraints():
    """the permutation as constraints of degree 7 over the gate's wires: symbolic state = linear forms {atom: coefficient} over the atoms
    ('b', i) = (in_i + rc_0,i)^7, ('w', k) = wire_k^7, 'one'; every S-box input of rounds 1 .. 29 is a wire constrained to its form"""
    RC, MDS, _ = _poseidon_tables()
    state = [{("b", i): 1} for i in range(12)]            # after round 0's S-boxes

    def mds(st):
        out = []
        for r in range(12):
            f = {}
            for c in range(12):
                m = MDS[r][c]
                for a, v in st[c].items():
                    f[a] = (f.get(a, 0) + m * v) % P
            out.append(f)
        return out
    cons = []                                               # (form, target wire)
    state = mds(state)
    for rnd in range(1, 30):
        full = rnd < 4 or rnd >= 26
        for i in range(12):
            state[i] = dict(state[i])
            state[i]["one"] = (state[i].get("one", 0) + RC[12 * rnd + i]) % P
        for i in (range(12) if full else (0,)):
            w = poseidon_sbox_wire(rnd, i)
            cons.append((state[i], w))
            state[i] = {("w", w): 1}
        state = mds(state)
    for i in range(12):
        cons.append((state[i], POS_OUT + i))
    return cons, RC


def circuit_recursion_shaped(num_wires=136, num_routed=80):
    assert num_wires >= 130 and num_routed >= 80
    n_ops = min(N_ARITH_OPS_MAX, num_routed // 4)
    prog, gates = [], []
    W, K, PIH = 0, 1, 2
    C0, C1 = 3, 4                                           # constant columns behind the three selectors

    def constraint(monos):
        monos = [(c, f) for c, f in monos if c % P]
        prog.append(len(monos))
        for coef, factors in monos:
            prog.extend([_i64(coef), len(factors)])
            for kind, idx in factors:
                prog.extend([kind, idx])
    # ---- group 0: selector column 0, gates 0 .. 4
    gates.append((0, 0, 0, 5, len(prog), 0))
    off = len(prog)
    for k in range(n_ops):
        constraint([(1, [(K, C0), (W, 4 * k), (W, 4 * k + 1)]), (1, [(K, C1), (W, 4 * k + 2)]), (-1, [(W, 4 * k + 3)])])
    gates.append((0, 1, 0, 5, off, n_ops))
    off = len(prog)
    constraint([(1 << i, [(W, 1 + i)]) for i in range(N_LIMBS)] + [(-1, [(W, 0)])])
    for i in range(N_LIMBS):
        constraint([(1, [(W, 1 + i), (W, 1 + i)]), (-1, [(W, 1 + i)])])
    gates.append((0, 2, 0, 5, off, 1 + N_LIMBS))
    off = len(prog)
    for i in range(4):
        constraint([(1, [(W, i)]), (-1, [(PIH, i)])])
    gates.append((0, 3, 0, 5, off, 4))
    off = len(prog)
    constraint([(1, [(W, 0)]), (-1, [(K, C0)])])
    constraint([(1, [(W, 1)]), (-1, [(K, C1)])])
    gates.append((0, 4, 0, 5, off, 2))
    # ---- group 1: selector column 1, gates 5 .. 7
    off = len(prog)
    for j in range(U32_OPS):
        b = U32_STRIDE * j
        constraint([(1, [(W, b), (W, b + 1)]), (1, [(W, b + 2)]), (-1, [(W, b + 3)]), (-(1 << 32), [(W, b + 4)])])
        for h in range(2):
            constraint([(4 ** i, [(W, b + 5 + U32_LIMBS * h + i)]) for i in range(U32_LIMBS)] + [(-1, [(W, b + 3 + h)])])
        for i in range(2 * U32_LIMBS):                      # limb (limb - 1)(limb - 2)(limb - 3)
            l = (W, b + 5 + i)
            constraint([(1, [l] * 4), (-6, [l] * 3), (11, [l] * 2), (-6, [l])])
    gates.append((1, 5, 5, 8, off, U32_OPS * (3 + 2 * U32_LIMBS)))
    off = len(prog)
    for j in range(RA_COPIES):
        b = RA_STRIDE * j
        idx, claimed, x, b0, b1 = (W, b), (W, b + 1), [(W, b + 2 + q) for q in range(4)], (W, b + 6), (W, b + 7)
        constraint([(1, [b0, b0]), (-1, [b0])])
        constraint([(1, [b1, b1]), (-1, [b1])])
        constraint([(1, [idx]), (-1, [b0]), (-2, [b1])])
        # m0 = x0 + b0 (x1 - x0), m1 = x2 + b0 (x3 - x2), claimed = m0 + b1 (m1 - m0)
        constraint([(1, [x[0]]), (-1, [b0, x[0]]), (1, [b0, x[1]]), (-1, [b1, x[0]]), (1, [b1, b0, x[0]]), (-1, [b1, b0, x[1]]),
                    (1, [b1, x[2]]), (-1, [b1, b0, x[2]]), (1, [b1, b0, x[3]]), (-1, [claimed])])
    gates.append((1, 6, 5, 8, off, 4 * RA_COPIES))
    off = len(prog)
    al0, al1 = (W, 0), (W, 1)
    for i in range(RED_K):
        p0, p1 = ((W, 2), (W, 3)) if i == 0 else ((W, 4 + RED_K + 2 * (i - 1)), (W, 5 + RED_K + 2 * (i - 1)))
        a0, a1 = (W, 4 + RED_K + 2 * i), (W, 5 + RED_K + 2 * i)
        constraint([(1, [p0, al0]), (EXT_W, [p1, al1]), (1, [(W, 4 + i)]), (-1, [a0])])
        constraint([(1, [p0, al1]), (1, [p1, al0]), (-1, [a1])])
    gates.append((1, 7, 5, 8, off, 2 * RED_K))
    # ---- group 2: selector column 2, gate 8
    off = len(prog)
    cons, RC = _poseidon_constraints()
    binom = [1, 7, 21, 35, 35, 21, 7, 1]
    for form, target in cons:
        monos, const = [], form.get("one", 0)
        for atom, coef in form.items():
            if atom == "one":
                continue
            if atom[0] == "w":
                monos.append((coef, [(W, atom[1])] * 7))
            else:                                            # (in_i + c)^7 = sum_k C(7, k) c^(7 - k) in_i^k
                i, c = atom[1], RC[atom[1]]
                const = (const + coef * pow(c, 7, P)) % P
                for k in range(1, 8):
                    monos.append((coef * binom[k] * pow(c, 7 - k, P), [(W, POS_IN + i)] * k))
        constraint(monos + [(const, []), (-1, [(W, target)])])
    gates.append((2, 8, 8, 9, off, len(cons)))
    return {"num_wires": num_wires, "num_routed": num_routed, "num_constants": 5, "num_selectors": 3, "n_arith_ops": n_ops, "gates": gates,
            "programs": np.array(prog, dtype=np.int64), "num_gate_constraints": max(g[5] for g in gates), "gate_names": RICH_GATE_NAMES, "rich": True}


def gl_mul_small(a, k):
    return gl_mul(a, np.uint64(k))


def poseidon_rows(inp):
    """the permutation over the columns of inp [12][m] (vectorised), returning (out [12][m], {wire: values}) with every S-box input wire"""
    RC, MDS, _ = _poseidon_tables()
    st = [inp[i].copy() for i in range(12)]
    wires = {}

    def mds(v):
        # the entries are below 2^6: the 32-bit halves of the state words are accumulated exactly in uint64 (twelve products of at most 38
        # bits) and each output is reduced ONCE: lo + 2^32 hi with hi < 2^42, 2^64 = 2^32 - 1 (five times faster than twelve field products)
        lo = [x & M32 for x in v]
        hi = [x >> np.uint64(32) for x in v]
        out = []
        for r in range(12):
            al = np.zeros_like(v[0])
            ah = np.zeros_like(v[0])
            for c in range(12):
                k = np.uint64(MDS[r][c])
                if k:
                    al = al + lo[c] * k
                    ah = ah + hi[c] * k
            # value = al + 2^32 (ah_lo + 2^32 ah_hi) = al + 2^32 ah_lo + (2^32 - 1) ah_hi, every summand below 2^64
            ah_lo, ah_hi = ah & M32, ah >> np.uint64(32)
            acc = gl_add(al % PP, ((ah_lo << np.uint64(32)) % PP))
            acc = gl_add(acc, (ah_hi << np.uint64(32)) - ah_hi)
            out.append(acc)
        return out
    for rnd in range(30):
        full = rnd < 4 or rnd >= 26
        st = [gl_add(st[i], np.uint64(RC[12 * rnd + i])) for i in range(12)]
        for i in (range(12) if full else (0,)):
            if rnd > 0:
                wires[poseidon_sbox_wire(rnd, i)] = st[i]
            st[i] = gl_pow7(st[i])
        st = mds(st)
    return st, wires


# ---- the gates' witness generators as data (include/sipp_hip.h sipp_plonk_generator; oracle/plonk.h orc_plonk_generator) -----------------
GEN_ARITHMETIC, GEN_BASE_SPLIT, GEN_CONSTANT, GEN_PUBLIC_INPUT, GEN_U32_MUL_ADD, GEN_RANDOM_ACCESS, GEN_REDUCING, GEN_POSEIDON = range(1, 9)


def generators(circ):
    """[(kind, selector_index, row, p0 .. p4)] for the recursion-shaped circuit: one generator per gate that has dependent wires"""
    assert circ.get("rich"), "generators are described for the recursion-shaped gate set"
    return [(GEN_ARITHMETIC, 0, 1, circ["n_arith_ops"], 3, 4, 0, 0),
            (GEN_BASE_SPLIT, 0, 2, N_LIMBS, 1, 0, 0, 0),
            (GEN_PUBLIC_INPUT, 0, 3, 0, 0, 0, 0, 0),
            (GEN_CONSTANT, 0, 4, 2, 3, 0, 0, 0),
            (GEN_U32_MUL_ADD, 1, 5, U32_OPS, U32_STRIDE, U32_LIMBS, 0, 0),
            (GEN_RANDOM_ACCESS, 1, 6, RA_COPIES, RA_STRIDE, 2, 0, 0),
            (GEN_REDUCING, 1, 7, RED_K, EXT_W, 0, 0, 0),
            (GEN_POSEIDON, 2, 8, POS_IN, POS_OUT, POS_SBOX, 0, 0)]


def generated_wires(circ):
    """gate index -> the wires its generator WRITES (everything else on a row of that gate is an input or a free cell)"""
    n_ops = circ["n_arith_ops"]
    u32 = [U32_STRIDE * j + k for j in range(U32_OPS) for k in range(3, U32_STRIDE)]
    ra = [RA_STRIDE * j + k for j in range(RA_COPIES) for k in (1, 6, 7)]
    return {1: [4 * k + 3 for k in range(n_ops)], 2: list(range(1, 1 + N_LIMBS)), 3: [0, 1, 2, 3], 4: [0, 1], 5: u32, 6: ra,
            7: list(range(4 + RED_K, 4 + 3 * RED_K)), 8: list(range(POS_OUT, POS_OUT + 12)) + list(range(POS_SBOX, POS_SBOX + 106))}


def blank_generated(circ, wires, gate, value=0, sched=None):
    """a copy of the wire table with every generated cell overwritten (what the caller holds BEFORE witness generation); with a
    chain_schedule() also the input cells that copy constraints feed from other rows' outputs"""
    out = wires.copy()
    for g, ws in generated_wires(circ).items():
        rows = np.flatnonzero(gate == g)
        for w in ws:
            out[w, rows] = np.uint64(value)
    if sched:
        out.reshape(-1)[sched["copy_dst"].astype(np.int64)] = np.uint64(value)
    return out


def gate_rows(n):
    """the gate of every row: the mix of a recursive verifier -- Poseidon 5 / 16, arithmetic 3 / 16, U32, random access and reducing
    2 / 16 each, base sum, constant; row 0 the public-input row, a few no-ops"""
    pattern = np.array([8, 1, 8, 5, 6, 8, 1, 7, 8, 2, 1, 5, 8, 6, 7, 4], dtype=np.int64)
    gate = pattern[np.arange(n) % 16]
    gate[0] = 3                                             # the public-input row
    gate[1::64] = 0                                         # a few no-ops: rows whose routed wires are all free
    return gate


def chain_schedule(log_n, chain_len):
    """COPY CONSTRAINTS FROM OUTPUTS TO INPUTS, and the order they force on the generators (what plonky2's generate_partial_witness finds
    with its work list; here fixed at circuit-build time): the Poseidon rows form hash chains of `chain_len` links (a row's inputs 0 .. 3
    are the outputs 0 .. 3 of the row `stride` Poseidon rows before it -- Merkle paths), the first multiplicand of arithmetic row t is
    output 4 of Poseidon row t, the first coefficient of reducing row t is the first output of arithmetic row t.
    -> dict: n_levels; row_level [N]; rows (all rows sorted by level, then gate) + level_offsets [n_levels + 1]; copy_src / copy_dst (cell =
    wire * N + row; the copy runs after its source's level) sorted by level + copy_offsets [n_levels + 1]"""
    n = 1 << log_n
    gate = gate_rows(n)
    ip, ia, idr = (np.flatnonzero(gate == g) for g in (8, 1, 7))
    stride = max(1, -(-len(ip) // chain_len))
    lp = np.arange(len(ip)) // stride
    la = lp[:len(ia)] + 1
    lr = la[:len(idr)] + 1
    row_level = np.zeros(n, dtype=np.uint32)
    row_level[ip], row_level[ia], row_level[idr] = lp, la, lr
    src, dst, lev = [], [], []
    j = np.arange(stride, len(ip))
    for q in range(4):
        src.append((POS_OUT + q) * n + ip[j - stride]); dst.append((POS_IN + q) * n + ip[j]); lev.append(lp[j - stride])
    t = np.arange(len(ia))
    src.append((POS_OUT + 4) * n + ip[t]); dst.append(0 * n + ia); lev.append(lp[t])
    t = np.arange(len(idr))
    src.append(3 * n + ia[t]); dst.append(4 * n + idr); lev.append(la[t])
    src, dst, lev = (np.concatenate(v).astype(np.int64) for v in (src, dst, lev))
    o = np.argsort(lev, kind="stable")
    src, dst, lev = src[o], dst[o], lev[o]
    n_levels = int(row_level.max()) + 1
    rows = np.lexsort((np.arange(n), gate, row_level)).astype(np.uint32)     # by level, then by gate: the rows of a family side by side
    level_offsets = np.searchsorted(row_level[rows], np.arange(n_levels + 1)).astype(np.uint32)
    copy_offsets = np.searchsorted(lev, np.arange(n_levels + 1)).astype(np.uint32)
    return {"n_levels": n_levels, "row_level": row_level, "rows": rows, "level_offsets": level_offsets, "copy_src": src.astype(np.uint64),
            "copy_dst": dst.astype(np.uint64), "copy_offsets": copy_offsets, "chain_len": chain_len}


def witness_recursion_shaped(circ, log_n, seed, pih, inputs_only=False, chain_len=0):
    """inputs_only: the INPUT cells and the circuit's constants / sigmas only -- what a caller holds before generate_partial_witness; the
    field arithmetic of the generators (arithmetic, reducing, Poseidon rows) is skipped and their cells keep random words (same random
    stream: every input cell equals the full witness').  chain_len > 0: the circuit of chain_schedule() -- outputs wired to inputs of
    other rows by copy constraints (2-cycles of the wire permutation), generated level by level"""
    rng = np.random.default_rng(seed)
    n = 1 << log_n
    Wn, R, n_ops = circ["num_wires"], circ["num_routed"], circ["n_arith_ops"]
    gate = gate_rows(n)
    wires = rand_field(rng, (Wn, n))
    c0, c1 = rand_field(rng, n), rand_field(rng, n)
    ix = {g: np.flatnonzero(gate == g) for g in range(9)}
    sched = chain_schedule(log_n, chain_len) if chain_len else None
    # free (routable) cells: arithmetic inputs, Poseidon inputs, the random-access lists, the reducing coefficients, every routed wire of a noop row
    free = np.zeros((R, n), dtype=bool)
    for k in range(n_ops):
        free[4 * k:4 * k + 3, ix[1]] = True
    free[POS_IN:POS_IN + 12, ix[8]] = True
    for j in range(RA_COPIES):
        free[RA_STRIDE * j + 2:RA_STRIDE * j + 6, ix[6]] = True
    free[4:4 + RED_K, ix[7]] = True
    free[:, ix[0]] = True
    if sched:                                               # a cell fed by another gate's output is not free
        free.reshape(-1)[sched["copy_dst"].astype(np.int64)] = False
    cells = np.flatnonzero(free.reshape(-1)).astype(np.int64)
    order = rng.permutation(cells)
    order = order[:len(order) - len(order) % 3].reshape(-1, 3)
    perm = np.arange(R * n, dtype=np.int64)
    perm[order[:, 0]], perm[order[:, 1]], perm[order[:, 2]] = order[:, 1], order[:, 2], order[:, 0]
    if sched:                                               # output cell <-> the input cell it feeds: a 2-cycle
        cs_, cd_ = sched["copy_src"].astype(np.int64), sched["copy_dst"].astype(np.int64)
        assert int(max(cs_.max(), cd_.max())) < R * n
        perm[cs_], perm[cd_] = cd_, cs_
    vals = rand_field(rng, order.shape[0])
    flat = wires[:R].reshape(-1)
    for q in range(3):
        flat[order[:, q]] = vals
    wires[:R] = flat.reshape(R, n)
    # ---- outputs, gate by gate (on the rows of that gate only)
    def gen_arith(r):
        for k in range(n_ops):
            wires[4 * k + 3, r] = gl_add(gl_mul(c0[r], gl_mul(wires[4 * k, r], wires[4 * k + 1, r])), gl_mul(c1[r], wires[4 * k + 2, r]))

    def gen_reducing(r):
        a0, a1 = wires[2, r], wires[3, r]
        al0, al1 = wires[0, r], wires[1, r]
        for i in range(RED_K):
            n0 = gl_add(gl_add(gl_mul(a0, al0), gl_mul_small(gl_mul(a1, al1), EXT_W)), wires[4 + i, r])
            n1 = gl_add(gl_mul(a0, al1), gl_mul(a1, al0))
            wires[4 + RED_K + 2 * i, r], wires[5 + RED_K + 2 * i, r] = n0, n1
            a0, a1 = n0, n1

    def gen_poseidon(r):
        out, sb = poseidon_rows(np.stack([wires[POS_IN + i, r] for i in range(12)]))
        for i in range(12):
            wires[POS_OUT + i, r] = out[i]
        for w, v in sb.items():
            wires[w, r] = v
    ia = ix[1]
    ibs = ix[2]
    bits = rng.integers(0, 2, size=(N_LIMBS, len(ibs)), dtype=np.uint64)
    total = np.zeros(len(ibs), dtype=np.uint64)
    for i in range(N_LIMBS):
        wires[1 + i, ibs] = bits[i]
        total = total + (bits[i] << np.uint64(i))
    wires[0, ibs] = total % PP
    for i in range(4):
        wires[i, 0] = np.uint64(int(pih[i]))
    ic = ix[4]
    wires[0, ic], wires[1, ic] = c0[ic], c1[ic]
    iu = ix[5]
    for j in range(U32_OPS):
        b = U32_STRIDE * j
        a_, b_, c_ = (rng.integers(0, 1 << 32, size=len(iu), dtype=np.uint64) for _ in range(3))
        full = a_ * b_ + c_                                  # < 2^64: exact in uint64
        lo, hi = full & M32, full >> np.uint64(32)
        wires[b, iu], wires[b + 1, iu], wires[b + 2, iu], wires[b + 3, iu], wires[b + 4, iu] = a_, b_, c_, lo, hi
        for h, v in enumerate((lo, hi)):
            for i in range(U32_LIMBS):
                wires[b + 5 + U32_LIMBS * h + i, iu] = (v >> np.uint64(2 * i)) & np.uint64(3)
    ir = ix[6]
    for j in range(RA_COPIES):
        b = RA_STRIDE * j
        idx = rng.integers(0, 4, size=len(ir), dtype=np.uint64)
        wires[b, ir] = idx
        wires[b + 6, ir], wires[b + 7, ir] = idx & np.uint64(1), idx >> np.uint64(1)
        items = np.stack([wires[b + 2 + q, ir] for q in range(4)])
        wires[b + 1, ir] = items[idx.astype(np.int64), np.arange(len(ir))]
    idr = ix[7]
    ip = ix[8]
    if not inputs_only and not sched:
        gen_arith(ia)
        gen_reducing(idr)
        gen_poseidon(ip)
    elif not inputs_only:                                   # level by level: generators of the level, then the copies its outputs feed
        rl, flat = sched["row_level"], wires.reshape(-1)
        co, cs_, cd_ = sched["copy_offsets"], sched["copy_src"].astype(np.int64), sched["copy_dst"].astype(np.int64)
        for lv in range(sched["n_levels"]):
            for rows_g, gen in ((ip, gen_poseidon), (ia, gen_arith), (idr, gen_reducing)):
                r = rows_g[rl[rows_g] == lv]
                if len(r):
                    gen(r)
            flat[cd_[co[lv]:co[lv + 1]]] = flat[cs_[co[lv]:co[lv + 1]]]
    # ---- constants: three selector columns, the gates' constants; sigmas
    sels = [np.where((gate >= lo) & (gate < hi), gate, UNUSED).astype(np.uint64) for lo, hi in ((0, 5), (5, 8), (8, 9))]
    pw = powers(root_of_unity(log_n), n)
    ks = np.array([pow(7, j, P) for j in range(R)], dtype=np.uint64)
    sig = np.empty((R, n), dtype=np.uint64)
    pm = perm.reshape(R, n)
    for j in range(R):
        sig[j] = gl_mul(ks[pm[j] >> log_n], pw[pm[j] & (n - 1)])
    cs = np.concatenate([np.stack(sels + [c0, c1]), sig]).astype(np.uint64)
    return np.ascontiguousarray(wires), np.ascontiguousarray(cs), gate


def witness(circ, log_n, seed, pih, inputs_only=False, chain_len=0):
    """wires [num_wires][N], constants_sigmas [4 + num_routed][N] (VALUES, natural row order) satisfying every gate and a random wire
    permutation over the gates' free input cells (cycles of three cells, constant on a cycle); pih = hash_no_pad(public inputs)"""
    if circ.get("rich"):
        return witness_recursion_shaped(circ, log_n, seed, pih, inputs_only, chain_len)
    assert not inputs_only and not chain_len
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
