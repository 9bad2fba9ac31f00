"""Second, independent reading of the GENERIC layer under the three SIPP STARKs, in big-integer Python:
Goldilocks + quadratic extension, Poseidon (naive form: no fast partial rounds), the overwrite-mode sponge, Merkle caps,
PolynomialBatch commitments, the duplex Challenger and FRI (commit phase, proof of work, query verification).

TEST INFRASTRUCTURE ONLY; PARITY UNPINNED.  The algorithms live in plonky2 / starky @ InternetMaximalism/plonky2 541e127
(reference Cargo.toml:21,24), which is not vendored under /root/reference; the reference reaches them only through
src/verifier_circuit.rs:133-135 (the three *_exp_circuit calls) and src/transcript_native.rs:27,57 (Poseidon).
This file was written from the published structure of plonky2 (field/src/fft.rs, hash/poseidon.rs, hash/hashing.rs,
hash/merkle_tree.rs, iop/challenger.rs, fri/{oracle,prover,verifier,reduction_strategies}.rs) WITHOUT consulting
oracle/*.c, deliberately in a different shape (recursive FFT, naive Poseidon, dictionaries instead of flat arrays),
so that tests/test_oracle_second_reading.py compares two readings, not two copies.  What it cannot pin is whether
both readings recall upstream correctly.

Only tests/ import this module.
"""
import os

P = 2**64 - 2**32 + 1
GEN = 7                                   # F::MULTIPLICATIVE_GROUP_GENERATOR = F::coset_shift()
POWER_OF_TWO_GENERATOR = 1753635133440165772   # generator of the 2^32 subgroup
TWO_ADICITY = 32
W = 7                                     # quadratic extension F[X] / (X^2 - 7)

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


# ---------------------------------------------------------------------------------------------------- field
def primitive_root_of_unity(n_log):
    assert n_log <= TWO_ADICITY
    return pow(POWER_OF_TWO_GENERATOR, 1 << (TWO_ADICITY - n_log), P)


def inv(a):
    return pow(a, P - 2, P)


class Ext(tuple):
    """a0 + a1 X, X^2 = 7"""
    __slots__ = ()

    def __new__(cls, a0, a1=0):
        return tuple.__new__(cls, (a0 % P, a1 % P))

    def __add__(self, o):
        o = ext(o)
        return Ext(self[0] + o[0], self[1] + o[1])

    __radd__ = __add__

    def __sub__(self, o):
        o = ext(o)
        return Ext(self[0] - o[0], self[1] - o[1])

    def __mul__(self, o):
        o = ext(o)
        return Ext(self[0] * o[0] + W * self[1] * o[1], self[0] * o[1] + self[1] * o[0])

    __rmul__ = __mul__

    def inverse(self):
        # (a0 - a1 X) / (a0^2 - 7 a1^2)
        d = inv((self[0] * self[0] - W * self[1] * self[1]) % P)
        return Ext(self[0] * d, -self[1] * d)

    def __pow__(self, e):
        r, b = Ext(1), self
        while e:
            if e & 1:
                r = r * b
            b = b * b
            e >>= 1
        return r


def ext(x):
    return x if isinstance(x, Ext) else Ext(x)


# ---------------------------------------------------------------------------------------------------- Poseidon
def _load_round_constants():
    vals = []
    for line in open(os.path.join(ROOT, "data", "poseidon_goldilocks_rc.txt")):
        line = line.split("#")[0]
        vals += [int(t, 16) for t in line.replace(",", " ").split()]
    assert len(vals) == 360
    return vals


ALL_ROUND_CONSTANTS = _load_round_constants()
MDS_CIRC = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]
MDS_DIAG = [8] + [0] * 11
WIDTH, RATE, HALF_FULL, PARTIAL = 12, 8, 4, 22


# row r of the matrix is the circulant shifted by r, plus the diagonal:  M[r][c] = CIRC[(c - r) mod 12] + [c = r] DIAG[r]
MDS_ROWS = [tuple(MDS_CIRC[(c - r) % WIDTH] + (MDS_DIAG[r] if c == r else 0) for c in range(WIDTH)) for r in range(WIDTH)]


def _mds(state):
    return [sum(map(int.__mul__, row, state)) % P for row in MDS_ROWS]


def poseidon(state):
    """hash/poseidon.rs `poseidon_naive`: 4 full rounds, 22 partial rounds (S-box on lane 0 only), 4 full rounds"""
    s = [x % P for x in state]
    assert len(s) == WIDTH
    rnd = 0
    for phase, count in (("full", HALF_FULL), ("partial", PARTIAL), ("full", HALF_FULL)):
        for _ in range(count):
            s = [(x + ALL_ROUND_CONSTANTS[WIDTH * rnd + i]) % P for i, x in enumerate(s)]
            if phase == "full":
                s = [pow(x, 7, P) for x in s]
            else:
                s[0] = pow(s[0], 7, P)
            s = _mds(s)
            rnd += 1
    return s


def hash_n_to_m_no_pad(inputs, m):
    state = [0] * WIDTH
    inputs = list(inputs)
    for i in range(0, len(inputs), RATE):
        chunk = inputs[i:i + RATE]
        state[:len(chunk)] = chunk           # overwrite mode
        state = poseidon(state)
    out = []
    while True:
        for x in state[:RATE]:
            out.append(x)
            if len(out) == m:
                return out
        state = poseidon(state)


def hash_no_pad(inputs):
    return hash_n_to_m_no_pad(inputs, 4)


def hash_or_noop(inputs):
    inputs = list(inputs)
    if len(inputs) <= 4:
        return inputs + [0] * (4 - len(inputs))
    return hash_no_pad(inputs)


def two_to_one(left, right):
    return poseidon(list(left) + list(right) + [0] * 4)[:4]


# ---------------------------------------------------------------------------------------------------- FFT
def reverse_bits(x, bits):
    return int(format(x, "0%db" % bits)[::-1], 2) if bits else 0


def reverse_index_bits(v):
    n = len(v)
    bits = n.bit_length() - 1
    return [v[reverse_bits(i, bits)] for i in range(n)]


def _fft_rec(c, w, mul, add, sub):
    n = len(c)
    if n == 1:
        return c
    w2 = w * w % P
    ev, od = _fft_rec(c[0::2], w2, mul, add, sub), _fft_rec(c[1::2], w2, mul, add, sub)
    out = [None] * n
    t = 1
    for i in range(n // 2):
        x = mul(od[i], t)
        out[i] = add(ev[i], x)
        out[i + n // 2] = sub(ev[i], x)
        t = t * w % P
    return out


def fft(coeffs):
    """values[i] = sum_j coeffs[j] w^(i j), natural order; elements are ints or Ext"""
    n = len(coeffs)
    w = primitive_root_of_unity(n.bit_length() - 1)
    if isinstance(coeffs[0], Ext):
        return _fft_rec(list(coeffs), w, lambda a, t: a * t, lambda a, b: a + b, lambda a, b: a - b)
    return _fft_rec([c % P for c in coeffs], w, lambda a, t: a * t % P, lambda a, b: (a + b) % P, lambda a, b: (a - b) % P)


def ifft(values):
    n = len(values)
    r = fft(values)
    ninv = inv(n)
    out = [r[0]] + r[:0:-1]                  # index i -> -i
    return [x * ninv if isinstance(x, Ext) else x * ninv % P for x in out]


def coset_fft(coeffs, shift):
    s, out = 1, []
    for c in coeffs:
        out.append(c * s if isinstance(c, Ext) else c * s % P)
        s = s * shift % P
    return fft(out)


def lde(coeffs, rate_bits):
    zero = Ext(0) if isinstance(coeffs[0], Ext) else 0
    return list(coeffs) + [zero] * (len(coeffs) * ((1 << rate_bits) - 1))


# ---------------------------------------------------------------------------------------------------- Merkle
class MerkleTree:
    """hash/merkle_tree.rs: leaves hashed with hash_or_noop, pairs with two_to_one, the top 2^cap_height digests are the cap"""

    def __init__(self, leaves, cap_height):
        n = len(leaves)
        self.log_n = n.bit_length() - 1
        assert 1 << self.log_n == n and cap_height <= self.log_n
        self.leaves = leaves
        self.cap_height = cap_height
        self.levels = [[hash_or_noop(l) for l in leaves]]
        while len(self.levels[-1]) > (1 << cap_height):
            prev = self.levels[-1]
            self.levels.append([two_to_one(prev[2 * i], prev[2 * i + 1]) for i in range(len(prev) // 2)])
        self.cap = self.levels[-1]

    def prove(self, index):
        sib = []
        for lvl in self.levels[:-1]:
            sib.append(lvl[index ^ 1])
            index >>= 1
        return sib


def verify_merkle_proof_to_cap(leaf, index, cap, siblings):
    cur = hash_or_noop(leaf)
    for s in siblings:
        cur = two_to_one(s, cur) if index & 1 else two_to_one(cur, s)
        index >>= 1
    return cur == list(cap[index])


# ---------------------------------------------------------------------------------------------------- PolynomialBatch
class PolynomialBatch:
    """fri/oracle.rs: from_values = ifft per column, then from_coeffs: coset LDE (shift 7), transpose,
    reverse_index_bits on the rows, Merkle tree with cap.  No blinding (starky passes `false`)."""

    def __init__(self, coeffs, rate_bits, cap_height):
        self.polynomials = coeffs
        lde_values = [coset_fft(lde(c, rate_bits), GEN) for c in coeffs]
        rows = [[col[i] for col in lde_values] for i in range(len(lde_values[0]))]
        self.tree = MerkleTree(reverse_index_bits(rows), cap_height)

    @classmethod
    def from_values(cls, values, rate_bits, cap_height):
        return cls([ifft(v) for v in values], rate_bits, cap_height)


# ---------------------------------------------------------------------------------------------------- Challenger
class Challenger:
    """iop/challenger.rs: duplex sponge.  Observing clears pending outputs; 8 buffered inputs trigger a duplexing;
    challenges pop from the END of the rate portion of the state."""

    def __init__(self):
        self.state = [0] * WIDTH
        self.input, self.output = [], []

    def copy(self):
        c = Challenger()
        c.state, c.input, c.output = list(self.state), list(self.input), list(self.output)
        return c

    def observe(self, e):
        self.output = []
        self.input.append(e % P)
        if len(self.input) == RATE:
            self.duplexing()

    def observe_many(self, es):
        for e in es:
            self.observe(e)

    def observe_ext(self, e):
        self.observe_many(ext(e))

    def observe_cap(self, cap):
        for h in cap:
            self.observe_many(h)

    def duplexing(self):
        for i, x in enumerate(self.input):
            self.state[i] = x
        self.input = []
        self.state = poseidon(self.state)
        self.output = list(self.state[:RATE])

    def get(self):
        if self.input or not self.output:
            self.duplexing()
        return self.output.pop()

    def get_n(self, n):
        return [self.get() for _ in range(n)]

    def get_ext(self):
        a = self.get_n(2)
        return Ext(a[0], a[1])

    def get_hash(self):
        return self.get_n(4)


# ---------------------------------------------------------------------------------------------------- FRI
def reduction_arity_bits(arity_bits, final_poly_bits, degree_bits, rate_bits, cap_height):
    """FriReductionStrategy::ConstantArityBits(arity_bits, final_poly_bits)"""
    out = []
    while degree_bits > final_poly_bits and degree_bits + rate_bits - arity_bits >= cap_height:
        out.append(arity_bits)
        degree_bits -= arity_bits
    return out


def pow_ok(response, pow_bits):
    return pow_bits == 0 or response >> (64 - pow_bits) == 0       # leading_zeros >= pow_bits (the order has 64 bits)


def pow_response(challenger, rule, witness):
    """rule 0: observe the witness, the next challenge is the response (fri/prover.rs fri_proof_of_work, 2023);
    rule 1: hash_no_pad(challenger.get_hash() || witness)[0] (the earlier form).  Advances the challenger as the prover does."""
    if rule == 1:
        return hash_no_pad(challenger.get_hash() + [witness])[0]
    challenger.observe(witness)
    return challenger.get()


def fri_committed_trees(coeffs, arities, rate_bits, cap_height, challenger):
    """fri/prover.rs: coeffs = lde_final_poly coefficients (Ext, length n << rate_bits, upper part zero)"""
    trees, shift = [], GEN
    values = coset_fft(coeffs, shift)
    for ab in arities:
        arity = 1 << ab
        vals = reverse_index_bits(values)
        leaves = [[x for e in vals[i:i + arity] for x in e] for i in range(0, len(vals), arity)]
        tree = MerkleTree(leaves, cap_height)
        challenger.observe_cap(tree.cap)
        trees.append(tree)
        beta = challenger.get_ext()
        folded = []
        for i in range(0, len(coeffs), arity):
            acc = Ext(0)
            for c in reversed(coeffs[i:i + arity]):
                acc = acc * beta + c
            folded.append(acc)
        coeffs = folded
        shift = pow(shift, arity, P)
        values = coset_fft(coeffs, shift)
    final = coeffs[:len(coeffs) >> rate_bits]
    for c in final:
        challenger.observe_ext(c)
    return trees, final


def grind(challenger, rule, pow_bits):
    """smallest witness (this repository's deterministic choice; upstream's rayon find_any may return any valid one)"""
    base = challenger.copy()
    cur = base.get_hash() if rule == 1 else None
    w = 0
    while True:
        if rule == 1:
            r = hash_no_pad(cur + [w])[0]
        else:
            c = base.copy()
            c.observe(w)
            r = c.get()
        if pow_ok(r, pow_bits):
            return w
        w += 1


def compute_evaluation(x, x_index_within_coset, arity_bits, evals, beta):
    """fri/verifier.rs: interpolate {(coset_start g^i, evals_rev[i])} and evaluate at beta (barycentric form)"""
    arity = 1 << arity_bits
    g = primitive_root_of_unity(arity_bits)
    evals = reverse_index_bits(list(evals))
    rev = reverse_bits(x_index_within_coset, arity_bits)
    coset_start = x * pow(g, arity - rev, P) % P
    pts = [coset_start * pow(g, i, P) % P for i in range(arity)]
    total = Ext(0)
    for i, (xi, yi) in enumerate(zip(pts, evals)):
        num, den = yi, 1
        for k, xk in enumerate(pts):
            if k != i:
                num = num * (beta - xk)
                den = den * (xi - xk) % P
        total = total + num * inv(den)
    return total


def eval_poly(coeffs, x):
    acc = Ext(0)
    for c in reversed(coeffs):
        acc = acc * x + c
    return acc


def fri_verify_query(x_index, n_log, rate_bits, arities, alpha, batches, initial_rows, steps, betas, final_poly):
    """One query round of fri/verifier.rs.  batches: [(point, [column values at x in opening order], [opened values])];
    steps: [evals (list of Ext) per FRI layer].  Returns None on success or a string naming the failed check."""
    log_m = n_log + rate_bits
    subgroup_x = GEN * pow(primitive_root_of_unity(log_m), reverse_bits(x_index, log_m), P) % P
    # fri_combine_initial
    total, count_prev = Ext(0), 0
    for point, evals_at_x, opened in batches:
        acc_x, acc_o = Ext(0), Ext(0)
        for v, o in zip(reversed(evals_at_x), reversed(opened)):
            acc_x = acc_x * alpha + v
            acc_o = acc_o * alpha + o
        total = total * (alpha ** len(evals_at_x)) + (acc_x - acc_o) * (Ext(subgroup_x) - point).inverse()
    old = total * subgroup_x                    # the final polynomial was multiplied by X (plonky2 PR 436)
    for ab, evals, beta in zip(arities, steps, betas):
        within = x_index & ((1 << ab) - 1)
        if evals[within] != old:
            return "fold consistency"
        old = compute_evaluation(subgroup_x, within, ab, evals, beta)
        subgroup_x = pow(subgroup_x, 1 << ab, P)
        x_index >>= ab
    if eval_poly(final_poly, Ext(subgroup_x)) != old:
        return "final polynomial"
    return None


# ---------------------------------------------------------------------------------------------------- PLONK wire permutation
# Second reading of plonk/prover.rs (wires_permutation_partial_products_and_zs), plonk/plonk_common.rs (quotient_chunk_products,
# partial_products_and_z_gx, check_partial_products, reduce_with_powers, eval_l_0, ZeroPolyOnCoset) and plonk/vanishing_poly.rs
# (the Z(1) = 1 and partial-product terms of eval_vanishing_poly), in the shape of the Rust code: row-major lists, chunks, zips.
def get_unique_coset_shifts(num_shifts):
    return [pow(GEN, j, P) for j in range(num_shifts)]


def quotient_chunk_products(quotient_values, max_degree):
    out = []
    for k in range(0, len(quotient_values), max_degree):
        acc = 1
        for q in quotient_values[k:k + max_degree]:
            acc = acc * q % P
        out.append(acc)
    return out


def partial_products_and_z_gx(z_x, chunk_products):
    res, acc = [], z_x
    for c in chunk_products:
        acc = acc * c % P
        res.append(acc)
    return res


def wires_permutation_partial_products_and_zs(wires, sigmas, beta, gamma, max_degree):
    """wires, sigmas: [num_routed][N] value columns.  Returns the columns [partial products ..., Z] of ONE challenge (plonky2 pops the
    last one and moves it to the front of the batch)."""
    num_routed, n = len(wires), len(wires[0])
    k_is = get_unique_coset_shifts(num_routed)
    w = primitive_root_of_unity(n.bit_length() - 1)
    rows, z_x, x = [], 1, 1
    for i in range(n):
        quotients = []
        for j in range(num_routed):
            num = (wires[j][i] + beta * k_is[j] % P * x + gamma) % P
            den = (wires[j][i] + beta * sigmas[j][i] + gamma) % P
            quotients.append(num * inv(den) % P)
        row = partial_products_and_z_gx(z_x, quotient_chunk_products(quotients, max_degree))
        z_x, row[-1] = row[-1], z_x                          # the last term is Z(g x): swapped for Z(x)
        rows.append(row)
        x = x * w % P
    return [[r[c] for r in rows] for c in range(len(rows[0]))]


def eval_l_0(n, x):
    """L_0(x) = (x^n - 1) / (n (x - 1)) at an extension point"""
    return (x ** n - ext(1)) * ((x - ext(1)) * ext(n % P)).inverse()


def check_partial_products(numerators, denominators, partials, z_x, z_gx, max_degree):
    accs = [z_x] + list(partials)
    nexts = list(partials) + [z_gx]
    out = []
    for k, (prev, nxt) in enumerate(zip(accs, nexts)):
        num = den = ext(1)
        for v in numerators[k * max_degree:(k + 1) * max_degree]:
            num = num * v
        for v in denominators[k * max_degree:(k + 1) * max_degree]:
            den = den * v
        out.append(prev * num - nxt * den)
    return out


def eval_vanishing_poly_permutation(n_log, x, wires, sigmas, zs, zs_next, partial_products, betas, gammas, alphas, max_degree,
                                    constraint_terms=()):
    """eval_vanishing_poly at the extension point x from opened values, reduced with the powers of every alpha:
    vanishing_z_1_terms ++ vanishing_partial_products_terms ++ constraint_terms (the gates' terms at x are the caller's: the circuit is not
    vendored; () = the permutation argument alone)"""
    num_routed, num_ch = len(wires), len(betas)
    num_prods = len(partial_products) // num_ch
    k_is = get_unique_coset_shifts(num_routed)
    l_0 = eval_l_0(1 << n_log, x)
    z1_terms, pp_terms = [], []
    for i in range(num_ch):
        z1_terms.append(l_0 * (zs[i] - ext(1)))
        numerators = [wires[j] + x * ext(betas[i] * k_is[j] % P) + ext(gammas[i]) for j in range(num_routed)]
        denominators = [wires[j] + sigmas[j] * ext(betas[i]) + ext(gammas[i]) for j in range(num_routed)]
        pp_terms += check_partial_products(numerators, denominators, partial_products[i * num_prods:(i + 1) * num_prods], zs[i], zs_next[i],
                                           max_degree)
    terms = z1_terms + pp_terms + list(constraint_terms)
    out = []
    for a in alphas:                                             # reduce_with_powers
        acc = ext(0)
        for t in reversed(terms):
            acc = acc * ext(a) + t
        out.append(acc)
    return out


# ---------------------------------------------------------------------------------------------------- gates as data
# Second reading of gates/selectors.rs (compute_filter, UNUSED_SELECTOR), gates/gate.rs (eval_filtered: a gate's constraints times its
# filter, ADDED into the circuit's shared constraint vector) and plonk/vanishing_poly.rs (evaluate_gate_constraints), for a gate set
# given as data (include/sipp_hip.h "gates as data"; oracle/plonk.h): gates = [(selector_index, row, group_lo, group_hi, prog_offset,
# num_constraints)], programs = flat integers -- per constraint n_mono, per monomial coef, n_factors, (kind, index) pairs; kind 0 = wire,
# 1 = constant column, 2 = public_inputs_hash word.
UNUSED_SELECTOR = (1 << 32) - 1


def compute_filter(row, group_range, s, many_selector):
    f = ext(1)
    for i in group_range:
        if i != row:
            f = f * (ext(i) - s)
    if many_selector:
        f = f * (ext(UNUSED_SELECTOR) - s)
    return f


def evaluate_gate_constraints(gates, programs, num_selectors, local_wires, local_constants, public_inputs_hash):
    """all arguments at ONE point (extension elements): -> the circuit's constraint vector (length = max over gates)"""
    num_gate_constraints = max(g[5] for g in gates)
    constraints = [ext(0)] * num_gate_constraints
    prog = [int(x) for x in programs]
    for (sel, row, lo, hi, off, ncons) in gates:
        filt = compute_filter(row, range(lo, hi), local_constants[sel], num_selectors > 1)
        w = off
        for j in range(ncons):
            n_mono = prog[w]
            w += 1
            acc = ext(0)
            for _ in range(n_mono):
                term = ext(prog[w] % P)
                nf = prog[w + 1]
                w += 2
                for _f in range(nf):
                    kind, idx = prog[w], prog[w + 1]
                    w += 2
                    term = term * (local_wires[idx] if kind == 0 else local_constants[idx] if kind == 1 else ext(public_inputs_hash[idx]))
                acc = acc + term
            constraints[j] = constraints[j] + filt * acc
    return constraints
