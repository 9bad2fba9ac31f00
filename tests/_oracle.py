"""ctypes loader for oracle/liboracle.so (the CPU restatement).  Test infrastructure:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
P = 2**64 - 2**32 + 1

u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_lib = None


def build(force=False):
    if os.environ.get("SIPP_ORACLE_ASAN"):      # scripts/run_asan.sh: the AddressSanitizer + UBSan build of the same sources
        subprocess.check_call(["make", "-C", ODIR, "-s", "asan"])
        return os.path.join(ODIR, "liboracle_asan.so")
    so = os.path.join(ODIR, "liboracle.so")
    srcs = [os.path.join(ODIR, f) for f in os.listdir(ODIR) if f.endswith((".c", ".h", ".inc"))]
    srcs.append(os.path.join(os.path.dirname(ODIR), "data", "air_tables.h"))    # the AIR specification, shared with the product
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", ODIR, "-s"])
    return so


def load():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    sig = {
        "orc_poseidon_permute": (None, [u64p]),
        "orc_hash_no_pad": (None, [u64p, C.c_size_t, u64p]),
        "orc_two_to_one": (None, [u64p, u64p, u64p]),
        "orc_fft": (None, [u64p, C.c_uint]),
        "orc_ifft": (None, [u64p, C.c_uint]),
        "orc_naive_dft": (None, [u64p, u64p, C.c_uint]),
        "orc_coset_lde": (None, [u64p, C.c_uint, C.c_uint, C.c_uint64, u64p]),
        "orc_batch_from_values": (C.c_void_p, [u64p, C.c_size_t, C.c_uint, C.c_uint, C.c_uint]),
        "orc_batch_from_coeffs": (C.c_void_p, [u64p, C.c_size_t, C.c_uint, C.c_uint, C.c_uint]),
        "orc_batch_free": (None, [C.c_void_p]),
        "orc_batch_cap": (C.POINTER(C.c_uint64), [C.c_void_p]),
        "orc_batch_leaves": (C.POINTER(C.c_uint64), [C.c_void_p]),
        "orc_batch_coeffs": (C.POINTER(C.c_uint64), [C.c_void_p]),
        "orc_batch_level": (C.POINTER(C.c_uint64), [C.c_void_p, C.c_uint, C.POINTER(C.c_size_t)]),
        "orc_set_num_threads": (None, [C.c_int]),
        "orc_get_max_threads": (C.c_int, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def set_num_threads(n):
    """OpenMP threads of the oracle's parallel regions; returns what the runtime will really use (OMP_NUM_THREADS set after the
    OpenMP runtime was loaded -- e.g. after `import torch` -- is ignored, this is not)"""
    L = load()
    L.orc_set_num_threads(int(n))
    return int(L.orc_get_max_threads())


def rand_field(rng, shape):
    """uniform canonical Goldilocks elements"""
    a = rng.integers(0, P, size=shape, dtype=np.uint64, endpoint=False)
    return np.ascontiguousarray(a)


def permute(state):
    s = np.ascontiguousarray(np.array(state, dtype=np.uint64))
    load().orc_poseidon_permute(s)
    return s


def hash_no_pad(vals):
    v = np.ascontiguousarray(np.array(vals, dtype=np.uint64))
    out = np.zeros(4, dtype=np.uint64)
    load().orc_hash_no_pad(v, len(v), out)
    return out


def two_to_one(l, r):
    out = np.zeros(4, dtype=np.uint64)
    load().orc_two_to_one(np.ascontiguousarray(np.array(l, dtype=np.uint64)),
                          np.ascontiguousarray(np.array(r, dtype=np.uint64)), out)
    return out


class Batch:
    """PolynomialBatch::from_values / from_coeffs restatement (column-major input [ncols][N])."""

    def __init__(self, arr, log_n, rate_bits=1, cap_height=4, from_coeffs=False):
        L = load()
        arr = np.ascontiguousarray(arr, dtype=np.uint64)
        self.ncols = arr.shape[0]
        self.log_n = log_n
        self.rate_bits = rate_bits
        assert arr.shape[1] == 1 << log_n
        fn = L.orc_batch_from_coeffs if from_coeffs else L.orc_batch_from_values
        self.h = fn(arr, self.ncols, log_n, rate_bits, cap_height)
        self.cap_height = min(cap_height, log_n + rate_bits)
        self.L = L

    def _arr(self, ptr, shape):
        n = int(np.prod(shape))
        return np.ctypeslib.as_array(ptr, shape=(n,)).reshape(shape).copy()

    @property
    def cap(self):
        return self._arr(self.L.orc_batch_cap(self.h), (1 << self.cap_height, 4))

    @property
    def leaves(self):
        return self._arr(self.L.orc_batch_leaves(self.h), (1 << (self.log_n + self.rate_bits), self.ncols))

    @property
    def coeffs(self):
        return self._arr(self.L.orc_batch_coeffs(self.h), (self.ncols, 1 << self.log_n))

    def level(self, lvl):
        n = C.c_size_t()
        p = self.L.orc_batch_level(self.h, lvl, C.byref(n))
        return self._arr(p, (n.value, 4))

    def __del__(self):
        try:
            self.L.orc_batch_free(self.h)
        except Exception:
            pass


# ---------------- AIR / trace (oracle/air.c) ----------------
class OrcAir(C.Structure):
    _fields_ = [("name", C.c_char_p), ("kind", C.c_int), ("table_bits", C.c_int), ("cells_per_limb", C.c_int),
                ("n_main", C.c_int), ("checked_base", C.c_int), ("n_checked", C.c_int), ("n_ops", C.c_int),
                ("n_constraints", C.c_int), ("n_aux", C.c_int), ("pi_per_io", C.c_int), ("n_gadgets", C.c_int),
                ("carry_limbs", C.c_int), ("prog", C.POINTER(C.c_int64)), ("prog_len", C.c_int),
                ("aux", C.POINTER(C.c_int32)), ("log_rows", C.c_int), ("hardened", C.c_int),
                ("n_vflag", C.c_int), ("rowprog", C.POINTER(C.c_int8)), ("n_fields", C.c_int), ("flagdef", C.POINTER(C.c_int16)),
                ("flagoff", C.POINTER(C.c_int16)), ("n_vconst", C.c_int), ("vconst", C.POINTER(C.c_int64)), ("vconst_field", C.c_int)]


class OrcTrace(C.Structure):
    _fields_ = [("air", C.POINTER(OrcAir)), ("log_n", C.c_uint), ("num_io", C.c_size_t), ("width", C.c_int),
                ("trace", C.POINTER(C.c_uint64)), ("pis", C.POINTER(C.c_uint32))]


class Trace:
    def __init__(self, kind, ios):
        L = load()
        L.orc_trace_build.restype = C.POINTER(OrcTrace)
        L.orc_trace_build.argtypes = [C.c_int, u32p, C.c_size_t, C.POINTER(C.c_int)]
        L.orc_trace_free.argtypes = [C.POINTER(OrcTrace)]
        L.orc_trace_check_row.restype = C.c_long
        L.orc_trace_check_row.argtypes = [C.POINTER(OrcTrace), C.c_size_t]
        ios = np.ascontiguousarray(ios, dtype=np.uint32)
        err = C.c_int()
        self.L = L
        self.p = L.orc_trace_build(kind, ios, ios.shape[0], C.byref(err))
        self.err = err.value
        if not self.p:
            raise RuntimeError("orc_trace_build failed: %d" % err.value)
        t = self.p.contents
        self.log_n, self.width, self.num_io = t.log_n, t.width, t.num_io
        self.air = t.air.contents

    def array(self):
        """view [width][N] (no copy)"""
        n = 1 << self.log_n
        return np.ctypeslib.as_array(self.p.contents.trace, shape=(self.width * n,)).reshape(self.width, n)

    def check_row(self, r):
        return self.L.orc_trace_check_row(self.p, r)

    def __del__(self):
        try:
            if self.p:
                self.L.orc_trace_free(self.p)
        except Exception:
            pass


# ---------------- the map Fp2 -> E'(Fp2) in front of the BLS example (oracle/mapg2.c) ----------------
def map_to_g2(u_words):
    """u: [n][16] u32 -> MapToG2 records [n][48] = (u, x, y), x / y by the C reading of the Shallue - van de Woestijne map"""
    L = load()
    L.orc_map_to_g2.argtypes = [u32p, u32p]
    u = np.ascontiguousarray(u_words, dtype=np.uint32).reshape(-1, 16)
    out = np.zeros((u.shape[0], 48), dtype=np.uint32)
    for i in range(u.shape[0]):
        xy = np.zeros(32, dtype=np.uint32)
        rc = L.orc_map_to_g2(np.ascontiguousarray(u[i]), xy)
        if rc != 0:
            raise RuntimeError("orc_map_to_g2 failed: %d" % rc)
        out[i, :16] = u[i]
        out[i, 16:] = xy
    return out


# ---------------- the final pairing of the BLS example (oracle/pairing.c) ----------------
def pairing(pq_words):
    """P | Q (48 u32) -> the 96 words of e(P, Q) (arkworks' value) by the C reading of the pairing AIR's schedule"""
    L = load()
    L.orc_pairing.argtypes = [u32p, u32p]
    out = np.zeros(96, dtype=np.uint32)
    rc = L.orc_pairing(np.ascontiguousarray(pq_words, dtype=np.uint32), out)
    if rc != 0:
        raise RuntimeError("orc_pairing failed: %d" % rc)
    return out


def pairing_record_ok(rec):
    L = load()
    L.orc_pairing_record_ok.argtypes = [u32p]
    return bool(L.orc_pairing_record_ok(np.ascontiguousarray(rec, dtype=np.uint32)))


# ---------------- STARK prover / verifier (oracle/stark.c) ----------------
class OrcConfig(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("rate_bits", "cap_height", "pow_bits", "arity_bits", "final_poly_bits",
                                          "num_queries", "num_challenges", "pow_rule", "fs_rule", "lookup_rule")]


def default_config():
    cfg = OrcConfig()
    load().orc_default_config(C.byref(cfg))
    return cfg


def stark_prove(kind, ios, cfg=None):
    L = load()
    cfg = cfg or default_config()
    ios = np.ascontiguousarray(ios, dtype=np.uint32)
    L.orc_stark_prove.argtypes = [C.c_int, u32p, C.c_size_t, C.POINTER(OrcConfig), C.POINTER(C.POINTER(C.c_uint64)),
                                  C.POINTER(C.c_size_t)]
    L.orc_free.argtypes = [C.c_void_p]
    out = C.POINTER(C.c_uint64)()
    n = C.c_size_t()
    rc = L.orc_stark_prove(kind, ios, ios.shape[0], C.byref(cfg), C.byref(out), C.byref(n))
    if rc != 0:
        raise RuntimeError("orc_stark_prove failed: %d" % rc)
    proof = np.ctypeslib.as_array(out, shape=(n.value,)).copy()
    L.orc_free(out)
    return proof


def stark_prove_trace(trace, cfg=None):
    """prove from a filled `Trace` (possibly tampered through Trace.array()): whatever the cells hold is committed"""
    L = load()
    cfg = cfg or default_config()
    L.orc_stark_prove_trace.argtypes = [C.POINTER(OrcTrace), C.POINTER(OrcConfig), C.POINTER(C.POINTER(C.c_uint64)),
                                        C.POINTER(C.c_size_t)]
    L.orc_free.argtypes = [C.c_void_p]
    out = C.POINTER(C.c_uint64)()
    n = C.c_size_t()
    rc = L.orc_stark_prove_trace(trace.p, C.byref(cfg), C.byref(out), C.byref(n))
    if rc != 0:
        raise RuntimeError("orc_stark_prove_trace failed: %d" % rc)
    proof = np.ctypeslib.as_array(out, shape=(n.value,)).copy()
    L.orc_free(out)
    return proof


def stark_verify(proof, cfg=None):
    L = load()
    cfg = cfg or default_config()
    proof = np.ascontiguousarray(proof, dtype=np.uint64)
    L.orc_stark_verify.argtypes = [u64p, C.c_size_t, C.POINTER(OrcConfig)]
    L.orc_stark_verify.restype = C.c_int
    return L.orc_stark_verify(proof, len(proof), C.byref(cfg))


# ---------------- generic opening proofs (oracle/fri.c) ----------------
class OrcFriParams(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("rate_bits", "cap_height", "pow_bits", "num_queries", "pow_rule", "hiding", "n_rounds")] + \
               [("arity_bits", C.c_uint32 * 32)]


def fri_params(rate_bits=1, cap_height=4, pow_bits=16, num_queries=84, pow_rule=0, hiding=0, arity_bits=4, final_poly_bits=5,
               degree_bits=None, arities=None):
    """FriParams: ConstantArityBits(arity_bits, final_poly_bits) for `degree_bits`, or an explicit list `arities`"""
    p = OrcFriParams()
    p.rate_bits, p.cap_height, p.pow_bits, p.num_queries, p.pow_rule, p.hiding = rate_bits, cap_height, pow_bits, num_queries, pow_rule, hiding
    if arities is not None:
        p.n_rounds = len(arities)
        for i, a in enumerate(arities):
            p.arity_bits[i] = a
    else:
        L = load()
        L.orc_fri_const_arity.argtypes = [C.POINTER(OrcFriParams), C.c_uint, C.c_uint, C.c_uint]
        L.orc_fri_const_arity(C.byref(p), arity_bits, final_poly_bits, degree_bits)
    return p


class OrcPolyRange(C.Structure):
    _fields_ = [("oracle", C.c_uint32), ("col_begin", C.c_uint32), ("col_end", C.c_uint32)]


class OrcFriBatch(C.Structure):
    _fields_ = [("point", C.c_uint64 * 2), ("n_ranges", C.c_uint32), ("ranges", C.POINTER(OrcPolyRange))]


class OrcChallenger(C.Structure):
    _fields_ = [("state", C.c_uint64 * 12), ("in_buf", C.c_uint64 * 8), ("n_in", C.c_size_t), ("out_buf", C.c_uint64 * 8),
                ("n_out", C.c_size_t)]


def challenger(observe=()):
    L = load()
    ch = OrcChallenger()
    L.orc_chal_init(C.byref(ch))
    L.orc_chal_observe.argtypes = [C.POINTER(OrcChallenger), C.c_uint64]
    for e in observe:
        L.orc_chal_observe(C.byref(ch), int(e))
    return ch


def make_batches(batches):
    """[(point (c0, c1), [(oracle, col_begin, col_end), ...]), ...] -> ctypes array (keeps the range arrays alive)"""
    arr = (OrcFriBatch * len(batches))()
    keep = []
    for i, (pt, ranges) in enumerate(batches):
        r = (OrcPolyRange * len(ranges))(*[OrcPolyRange(*x) for x in ranges])
        keep.append(r)
        arr[i].point[0], arr[i].point[1] = int(pt[0]), int(pt[1])
        arr[i].n_ranges = len(ranges)
        arr[i].ranges = r
    arr._keep = keep
    return arr


class SaltedBatch:
    """orc_batch_salted: PolynomialBatch with `blinding` -- SALT_SIZE extra words per leaf, supplied by the caller"""

    def __init__(self, data, log_n, rate_bits, cap_height, from_values=True, salt=None):
        L = load()
        L.orc_batch_salted.restype = C.c_void_p
        L.orc_batch_salted.argtypes = [u64p, C.c_int, C.c_size_t, C.c_uint, C.c_uint, C.c_uint, C.c_void_p, C.c_size_t]
        data = np.ascontiguousarray(data, dtype=np.uint64)
        self.ncols, self.log_n, self.rate_bits = data.shape[0], log_n, rate_bits
        self.n_salt = 0 if salt is None else salt.shape[0]
        self.salt = None if salt is None else np.ascontiguousarray(salt, dtype=np.uint64)
        self.h = L.orc_batch_salted(data, int(from_values), self.ncols, log_n, rate_bits, cap_height,
                                    None if salt is None else self.salt.ctypes.data, self.n_salt)
        self.cap_height = min(cap_height, log_n + rate_bits)
        self.L = L

    @property
    def cap(self):
        return np.ctypeslib.as_array(self.L.orc_batch_cap(self.h), shape=((1 << self.cap_height) * 4,)).reshape(-1, 4).copy()

    @property
    def leaves(self):
        ll = self.ncols + self.n_salt
        m = 1 << (self.log_n + self.rate_bits)
        return np.ctypeslib.as_array(self.L.orc_batch_leaves(self.h), shape=(m * ll,)).reshape(m, ll).copy()

    @property
    def coeffs(self):
        n = 1 << self.log_n
        return np.ctypeslib.as_array(self.L.orc_batch_coeffs(self.h), shape=(self.ncols * n,)).reshape(self.ncols, n).copy()

    def __del__(self):
        try:
            self.L.orc_batch_free(self.h)
        except Exception:
            pass


def fri_prove_openings(oracles, batches, log_n, params, ch):
    L = load()
    L.orc_fri_prove_openings.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.POINTER(OrcFriBatch), C.c_size_t, C.c_uint,
                                         C.POINTER(OrcFriParams), C.POINTER(OrcChallenger), C.POINTER(C.POINTER(C.c_uint64)),
                                         C.POINTER(C.c_size_t)]
    L.orc_free.argtypes = [C.c_void_p]
    hs = (C.c_void_p * len(oracles))(*[o.h for o in oracles])
    barr = make_batches(batches)
    out = C.POINTER(C.c_uint64)()
    n = C.c_size_t()
    rc = L.orc_fri_prove_openings(hs, len(oracles), barr, len(batches), log_n, C.byref(params), C.byref(ch), C.byref(out), C.byref(n))
    if rc != 0:
        raise RuntimeError("orc_fri_prove_openings failed: %d" % rc)
    proof = np.ctypeslib.as_array(out, shape=(n.value,)).copy()
    L.orc_free(out)
    return proof


def fri_verify_openings(proof, caps, ncols, n_salt, batches, log_n, params, ch):
    L = load()
    L.orc_fri_verify_openings.argtypes = [u64p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_size_t,
                                          C.POINTER(OrcFriBatch), C.c_size_t, C.c_uint, C.POINTER(OrcFriParams),
                                          C.POINTER(OrcChallenger)]
    L.orc_fri_verify_openings.restype = C.c_int
    caps = [np.ascontiguousarray(c, dtype=np.uint64) for c in caps]
    cp = (C.c_void_p * len(caps))(*[c.ctypes.data for c in caps])
    nc = (C.c_int * len(caps))(*ncols)
    ns = (C.c_int * len(caps))(*n_salt)
    barr = make_batches(batches)
    proof = np.ascontiguousarray(proof, dtype=np.uint64)
    return L.orc_fri_verify_openings(proof, len(proof), cp, nc, ns, len(caps), barr, len(batches), log_n, C.byref(params), C.byref(ch))


# ---------------- plonky2's wire permutation argument (oracle/plonk.c) ----------------
class OrcPlonkParams(C.Structure):
    _fields_ = [("num_routed_wires", C.c_uint32), ("max_degree", C.c_uint32), ("num_challenges", C.c_uint32)]


def plonk_params(num_routed_wires=80, max_degree=8, num_challenges=2):
    return OrcPlonkParams(num_routed_wires, max_degree, num_challenges)


def plonk_num_prods(p):
    return (p.num_routed_wires + p.max_degree - 1) // p.max_degree - 1


def plonk_zs_cols(p):
    return p.num_challenges * (1 + plonk_num_prods(p))


def _plonk_lib():
    L = load()
    if not getattr(L, "_plonk_sigs", False):
        pp = C.POINTER(OrcPlonkParams)
        L.orc_plonk_sigmas_from_perm.argtypes = [u32p, C.c_uint32, C.c_uint, u64p]
        L.orc_plonk_zs_partial_products.argtypes = [u64p, u64p, C.c_uint, pp, u64p, u64p, u64p]
        L.orc_plonk_quotient_chunks.argtypes = [u64p, u64p, u64p, C.c_uint, pp, u64p, u64p, u64p, u64p]
        L.orc_plonk_quotient_chunks.restype = C.c_int
        L.orc_plonk_perm_prove.argtypes = [u64p, u64p, C.c_uint, pp, C.POINTER(OrcFriParams), u64p, u64p, C.POINTER(C.POINTER(C.c_uint64)),
                                           C.POINTER(C.c_size_t)]
        L.orc_plonk_perm_prove.restype = C.c_int
        L.orc_plonk_perm_verify.argtypes = [u64p, C.c_size_t, u64p, pp, C.POINTER(OrcFriParams), u64p, u64p]
        L.orc_plonk_perm_verify.restype = C.c_int
        L._plonk_sigs = True
    return L


def plonk_random_instance(seed, log_n, num_routed, n_cycles=None):
    """wires [R][N] that satisfy the copy constraints of a random wire permutation, and that permutation's sigma VALUES:
    positions (column j, row i) <-> index j N + i; the permutation is a product of random cycles, wires are constant on cycles"""
    rng = np.random.default_rng(seed)
    n = 1 << log_n
    total = num_routed * n
    order = rng.permutation(total).astype(np.uint32)
    n_cycles = n_cycles or max(1, total // 3)
    cuts = np.sort(rng.choice(np.arange(1, total), size=min(n_cycles - 1, total - 1), replace=False)) if total > 1 else np.array([], dtype=int)
    perm = np.empty(total, dtype=np.uint32)
    vals = np.empty(total, dtype=np.uint64)
    start = 0
    for end in list(cuts) + [total]:
        cyc = order[start:end]
        perm[cyc] = np.roll(cyc, -1)                       # sigma maps each position to the next one of its cycle
        vals[cyc] = rand_field(rng, (1,))[0]
        start = end
    sig = np.zeros((num_routed, n), dtype=np.uint64)
    _plonk_lib().orc_plonk_sigmas_from_perm(perm, num_routed, log_n, sig.reshape(-1))
    return vals.reshape(num_routed, n).copy(), sig, perm


class OrcPlonkGates(C.Structure):
    _fields_ = [("num_mul", C.c_uint32)]


def _plonk_gate_sigs(L):
    if not getattr(L, "_plonk_gate_sigs", False):
        pp, gp = C.POINTER(OrcPlonkParams), C.POINTER(OrcPlonkGates)
        L.orc_plonk_gate_terms_coset.argtypes = [u64p, C.c_uint, C.c_uint, gp, u64p]
        L.orc_plonk_gate_terms_coset.restype = None
        L.orc_plonk_quotient_chunks_ex.argtypes = [u64p, u64p, u64p, C.c_uint, pp, u64p, u64p, u64p, u64p, C.c_uint32, u64p]
        L.orc_plonk_quotient_chunks_ex.restype = C.c_int
        L.orc_plonk_prove_ex.argtypes = [u64p, u64p, C.c_uint, pp, C.POINTER(OrcFriParams), u64p, u64p, C.c_uint32, gp,
                                         C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_size_t)]
        L.orc_plonk_prove_ex.restype = C.c_int
        L.orc_plonk_verify_ex.argtypes = [u64p, C.c_size_t, u64p, pp, C.POINTER(OrcFriParams), u64p, gp]
        L.orc_plonk_verify_ex.restype = C.c_int
        L._plonk_gate_sigs = True
    return L


def plonk_gate_instance(seed, log_n, num_routed, num_mul):
    """the synthetic circuit of oracle/plonk.h: num_mul product gates w_{3k} w_{3k+1} = w_{3k+2} on EVERY row, and a random wire
    permutation over all the other columns (cycles with constant values); the output columns 3k + 2 are unrouted in effect (sigma =
    identity there).  Returns wires, sigma values, the permutation."""
    rng = np.random.default_rng(seed)
    n = 1 << log_n
    assert 3 * num_mul <= num_routed
    outs = set(3 * k + 2 for k in range(num_mul))
    free_pos = np.array([j * n + i for j in range(num_routed) if j not in outs for i in range(n)], dtype=np.uint32)
    total = num_routed * n
    perm = np.arange(total, dtype=np.uint32)
    vals = np.zeros(total, dtype=np.uint64)
    order = rng.permutation(free_pos)
    cuts = np.sort(rng.choice(np.arange(1, len(order)), size=max(1, len(order) // 3) - 1, replace=False))
    start = 0
    for end in list(cuts) + [len(order)]:
        cyc = order[start:end]
        perm[cyc] = np.roll(cyc, -1)
        vals[cyc] = rand_field(rng, (1,))[0]
        start = end
    w = vals.reshape(num_routed, n).copy()
    for k in range(num_mul):
        a, b = w[3 * k].astype(object), w[3 * k + 1].astype(object)
        w[3 * k + 2] = np.array([int(x) * int(y) % P for x, y in zip(a, b)], dtype=np.uint64)
    sig = np.zeros((num_routed, n), dtype=np.uint64)
    _plonk_lib().orc_plonk_sigmas_from_perm(perm, num_routed, log_n, sig.reshape(-1))
    return w, sig, perm


def plonk_gate_terms_coset(wires_c, log_n, log_d, num_mul):
    """term k = w_{3k} w_{3k+1} - w_{3k+2} on the coset 7 <w_{N D}>, natural order: [num_mul][N D]"""
    L = _plonk_gate_sigs(_plonk_lib())
    out = np.zeros((num_mul, (1 << log_n) << log_d), dtype=np.uint64)
    g = OrcPlonkGates(num_mul)
    L.orc_plonk_gate_terms_coset(np.ascontiguousarray(wires_c).reshape(-1), log_n, log_d, C.byref(g), out.reshape(-1))
    return out


def plonk_quotient_chunks_ex(wires_c, sigmas_c, zs_c, log_n, p, betas, gammas, alphas, gate_terms):
    L = _plonk_gate_sigs(_plonk_lib())
    out = np.zeros((p.num_challenges * p.max_degree, 1 << log_n), dtype=np.uint64)
    gt = np.ascontiguousarray(gate_terms, dtype=np.uint64)
    rc = L.orc_plonk_quotient_chunks_ex(np.ascontiguousarray(wires_c).reshape(-1), np.ascontiguousarray(sigmas_c).reshape(-1),
                                        np.ascontiguousarray(zs_c).reshape(-1), log_n, C.byref(p), np.asarray(betas, dtype=np.uint64),
                                        np.asarray(gammas, dtype=np.uint64), np.asarray(alphas, dtype=np.uint64), gt.reshape(-1), gt.shape[0],
                                        out.reshape(-1))
    if rc:
        raise RuntimeError("orc_plonk_quotient_chunks_ex: %d" % rc)
    return out


def plonk_prove_ex(wires, sigmas, log_n, p, fp, digest, public_inputs, num_mul):
    L = _plonk_gate_sigs(_plonk_lib())
    out = C.POINTER(C.c_uint64)()
    n = C.c_size_t()
    pis = np.asarray(list(public_inputs) or [0], dtype=np.uint64)
    g = OrcPlonkGates(num_mul)
    rc = L.orc_plonk_prove_ex(np.ascontiguousarray(wires).reshape(-1), np.ascontiguousarray(sigmas).reshape(-1), log_n, C.byref(p), C.byref(fp),
                              np.asarray(digest, dtype=np.uint64), pis, len(list(public_inputs)), C.byref(g), C.byref(out), C.byref(n))
    if rc:
        raise RuntimeError("orc_plonk_prove_ex: %d" % rc)
    pf = np.ctypeslib.as_array(out, shape=(n.value,)).copy()
    L.orc_free.argtypes = [C.c_void_p]
    L.orc_free(out)
    return pf


def plonk_verify_ex(proof, sigmas_cap, p, fp, digest, num_mul):
    L = _plonk_gate_sigs(_plonk_lib())
    proof = np.ascontiguousarray(proof, dtype=np.uint64)
    g = OrcPlonkGates(num_mul)
    return L.orc_plonk_verify_ex(proof, len(proof), np.ascontiguousarray(sigmas_cap, dtype=np.uint64).reshape(-1), C.byref(p), C.byref(fp),
                                 np.asarray(digest, dtype=np.uint64), C.byref(g))


# ---- gates as data (oracle/plonk_gates.c) ----
class OrcPlonkGate(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("selector_index", "row", "group_lo", "group_hi", "prog_offset", "num_constraints")]


class OrcPlonkCircuit(C.Structure):
    _fields_ = [("num_wires", C.c_uint32), ("num_constants", C.c_uint32), ("num_selectors", C.c_uint32), ("num_gates", C.c_uint32),
                ("gates", C.POINTER(OrcPlonkGate)), ("programs", C.POINTER(C.c_int64)), ("program_words", C.c_uint32)]


def plonk_circuit(circ, cls=(OrcPlonkGate, OrcPlonkCircuit)):
    """tools/plonk_synth.circuit() -> the C struct (keeps the arrays it points into alive as attributes)"""
    G, CC = cls
    gates = (G * len(circ["gates"]))(*[G(*[int(x) for x in g]) for g in circ["gates"]])
    prog = np.ascontiguousarray(circ["programs"], dtype=np.int64)
    c = CC(circ["num_wires"], circ["num_constants"], circ["num_selectors"], len(circ["gates"]), gates,
           prog.ctypes.data_as(C.POINTER(C.c_int64)), len(prog))
    c._keep = (gates, prog)
    return c


def _plonk_gates_lib():
    L = _plonk_lib()
    if not getattr(L, "_plonk_gates_sigs", False):
        pp, cp, fpp = C.POINTER(OrcPlonkParams), C.POINTER(OrcPlonkCircuit), C.POINTER(OrcFriParams)
        L.orc_plonk_prove_gates.argtypes = [u64p, u64p, C.c_uint, pp, fpp, cp, u64p, u64p, C.c_uint32, C.POINTER(C.POINTER(C.c_uint64)),
                                            C.POINTER(C.c_size_t)]
        L.orc_plonk_prove_gates.restype = C.c_int
        L.orc_plonk_verify_gates.argtypes = [u64p, C.c_size_t, u64p, pp, fpp, cp, u64p]
        L.orc_plonk_verify_gates.restype = C.c_int
        L.orc_plonk_gate_constraints_base.argtypes = [cp, u64p, u64p, u64p, u64p]
        L.orc_plonk_gate_constraints_base.restype = None
        L.orc_plonk_circuit_check.argtypes = [cp, pp]
        L.orc_plonk_circuit_check.restype = C.c_int
        L._plonk_gates_sigs = True
    return L


def plonk_prove_gates(wires, constants_sigmas, log_n, p, fp, circ, digest, public_inputs):
    L = _plonk_gates_lib()
    cc = plonk_circuit(circ)
    out = C.POINTER(C.c_uint64)()
    n = C.c_size_t()
    pis = np.asarray(list(public_inputs) or [0], dtype=np.uint64)
    rc = L.orc_plonk_prove_gates(np.ascontiguousarray(wires, dtype=np.uint64).reshape(-1), np.ascontiguousarray(constants_sigmas, dtype=np.uint64).reshape(-1),
                                 log_n, C.byref(p), C.byref(fp), C.byref(cc), np.asarray(digest, dtype=np.uint64), pis, len(list(public_inputs)),
                                 C.byref(out), C.byref(n))
    if rc:
        raise RuntimeError("orc_plonk_prove_gates: %d" % rc)
    pf = np.ctypeslib.as_array(out, shape=(n.value,)).copy()
    L.orc_free.argtypes = [C.c_void_p]
    L.orc_free(out)
    return pf


def plonk_verify_gates(proof, cs_cap, p, fp, circ, digest):
    L = _plonk_gates_lib()
    cc = plonk_circuit(circ)
    proof = np.ascontiguousarray(proof, dtype=np.uint64)
    return L.orc_plonk_verify_gates(proof, len(proof), np.ascontiguousarray(cs_cap, dtype=np.uint64).reshape(-1), C.byref(p), C.byref(fp), C.byref(cc),
                                    np.asarray(digest, dtype=np.uint64))


# ---- the gates' witness generators (oracle/plonk_witness.c) ----
class OrcPlonkGenerator(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("selector_index", C.c_uint32), ("row", C.c_uint32), ("p", C.c_uint32 * 5)]


def plonk_generate_witness(wires, consts, log_n, gens, pih, threads=None):
    """orc_plonk_generate_witness on a COPY of wires [num_wires][N]; consts [num_constants][N]; gens = tools/plonk_synth.generators(circ)"""
    L = _plonk_lib()
    L.orc_plonk_generate_witness.argtypes = [u64p, u64p, C.c_uint, C.c_uint32, C.c_uint32, C.POINTER(OrcPlonkGenerator), C.c_size_t, u64p]
    L.orc_plonk_generate_witness.restype = C.c_int
    w = np.ascontiguousarray(wires, dtype=np.uint64).copy()
    k = np.ascontiguousarray(consts, dtype=np.uint64)
    arr = (OrcPlonkGenerator * len(gens))(*[OrcPlonkGenerator(g[0], g[1], g[2], (C.c_uint32 * 5)(*[int(x) for x in g[3:8]])) for g in gens])
    rc = L.orc_plonk_generate_witness(w.reshape(-1), k.reshape(-1), log_n, w.shape[0], k.shape[0], arr, len(gens),
                                      np.asarray([int(x) for x in pih], dtype=np.uint64))
    if rc:
        raise RuntimeError("orc_plonk_generate_witness: %d" % rc)
    return w


def plonk_generate_witness_levels(wires, consts, log_n, gens, pih, sched):
    """orc_plonk_generate_witness_levels on a COPY of wires; sched = tools/plonk_synth.chain_schedule(log_n, chain_len)"""
    L = _plonk_lib()
    u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
    L.orc_plonk_generate_witness_levels.argtypes = [u64p, u64p, C.c_uint, C.c_uint32, C.c_uint32, C.POINTER(OrcPlonkGenerator), C.c_size_t, u64p,
                                                    C.c_uint32, u32p, u32p, u64p, u64p, u32p]
    L.orc_plonk_generate_witness_levels.restype = C.c_int
    w = np.ascontiguousarray(wires, dtype=np.uint64).copy()
    k = np.ascontiguousarray(consts, dtype=np.uint64)
    arr = (OrcPlonkGenerator * len(gens))(*[OrcPlonkGenerator(g[0], g[1], g[2], (C.c_uint32 * 5)(*[int(x) for x in g[3:8]])) for g in gens])
    a32 = lambda v: np.ascontiguousarray(v, dtype=np.uint32)
    a64 = lambda v: np.ascontiguousarray(v if len(v) else [0], dtype=np.uint64)
    rc = L.orc_plonk_generate_witness_levels(w.reshape(-1), k.reshape(-1), log_n, w.shape[0], k.shape[0], arr, len(gens),
                                             np.asarray([int(x) for x in pih], dtype=np.uint64), int(sched["n_levels"]), a32(sched["rows"]),
                                             a32(sched["level_offsets"]), a64(sched["copy_src"]), a64(sched["copy_dst"]), a32(sched["copy_offsets"]))
    if rc:
        raise RuntimeError("orc_plonk_generate_witness_levels: %d" % rc)
    return w


def plonk_gate_constraints_base(circ, wires_row, consts_row, pih):
    L = _plonk_gates_lib()
    cc = plonk_circuit(circ)
    out = np.zeros(max(1, circ["num_gate_constraints"]), dtype=np.uint64)
    L.orc_plonk_gate_constraints_base(C.byref(cc), np.ascontiguousarray(wires_row, dtype=np.uint64), np.ascontiguousarray(consts_row, dtype=np.uint64),
                                      np.asarray(pih, dtype=np.uint64), out)
    return out


def plonk_zs(wires, sigmas, log_n, p, betas, gammas):
    out = np.zeros((plonk_zs_cols(p), 1 << log_n), dtype=np.uint64)
    _plonk_lib().orc_plonk_zs_partial_products(np.ascontiguousarray(wires).reshape(-1), np.ascontiguousarray(sigmas).reshape(-1), log_n, C.byref(p),
                                               np.asarray(betas, dtype=np.uint64), np.asarray(gammas, dtype=np.uint64), out.reshape(-1))
    return out


def plonk_quotient_chunks(wires_c, sigmas_c, zs_c, log_n, p, betas, gammas, alphas):
    out = np.zeros((p.num_challenges * p.max_degree, 1 << log_n), dtype=np.uint64)
    rc = _plonk_lib().orc_plonk_quotient_chunks(np.ascontiguousarray(wires_c).reshape(-1), np.ascontiguousarray(sigmas_c).reshape(-1),
                                                np.ascontiguousarray(zs_c).reshape(-1), log_n, C.byref(p), np.asarray(betas, dtype=np.uint64),
                                                np.asarray(gammas, dtype=np.uint64), np.asarray(alphas, dtype=np.uint64), out.reshape(-1))
    if rc:
        raise RuntimeError("orc_plonk_quotient_chunks: %d" % rc)
    return out


def plonk_perm_prove(wires, sigmas, log_n, p, fp, digest=(1, 2, 3, 4), pih=(0, 0, 0, 0)):
    L = _plonk_lib()
    out = C.POINTER(C.c_uint64)()
    n = C.c_size_t()
    rc = L.orc_plonk_perm_prove(np.ascontiguousarray(wires).reshape(-1), np.ascontiguousarray(sigmas).reshape(-1), log_n, C.byref(p), C.byref(fp),
                                np.asarray(digest, dtype=np.uint64), np.asarray(pih, dtype=np.uint64), C.byref(out), C.byref(n))
    if rc:
        raise RuntimeError("orc_plonk_perm_prove: %d" % rc)
    pf = np.ctypeslib.as_array(out, shape=(n.value,)).copy()
    L.orc_free.argtypes = [C.c_void_p]
    L.orc_free(out)
    return pf


def plonk_perm_verify(proof, sigmas_cap, p, fp, digest=(1, 2, 3, 4), pih=(0, 0, 0, 0)):
    proof = np.ascontiguousarray(proof, dtype=np.uint64)
    return _plonk_lib().orc_plonk_perm_verify(proof, len(proof), np.ascontiguousarray(sigmas_cap, dtype=np.uint64).reshape(-1), C.byref(p), C.byref(fp),
                                              np.asarray(digest, dtype=np.uint64), np.asarray(pih, dtype=np.uint64))
