"""oracle/py/sipp_native.py -- restatement of the reference's native SIPP prover / verifier / transcript.

TEST INFRASTRUCTURE ONLY.  Follows, line by line in behaviour:
  Transcript            reference src/transcript_native.rs:14-77
  inner_product         reference src/prover_native.rs:15-23
  sipp_prove_native     reference src/prover_native.rs:26-80
  sipp_verify_native    reference src/verifier_native.rs:14-85
  statement limb layout reference src/statements.rs:24-39,134-169
  STARK obligation lists (x, offset, exp_val, output per IO)   reference src/verifier_circuit.rs:68-135
PARITY UNPINNED: pairing, Poseidon and MyFq12 come from un-vendored crates; the
Fq12 coefficient order assumed is ascending powers of w.
"""
import ctypes as C
import os

import numpy as np

from . import bn254 as bn

_ODIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = None


def _oracle():
    global _lib
    if _lib is None:
        _lib = C.CDLL(os.path.join(_ODIR, "liboracle.so"))
        _lib.orc_hash_no_pad.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        _lib.orc_hash_no_pad.restype = None
    return _lib


def hash_no_pad(vals):
    v = np.ascontiguousarray(np.array(vals, dtype=np.uint64))
    out = np.zeros(4, dtype=np.uint64)
    _oracle().orc_hash_no_pad(v.ctypes.data, len(v), out.ctypes.data)
    return [int(x) for x in out]


class Transcript:
    """src/transcript_native.rs:14-66"""

    def __init__(self):
        self.state = [0, 0, 0, 0]

    def append(self, msg):                       # :23-30  state <- H(state || msg)
        self.state = hash_no_pad(self.state + list(msg))

    def append_fq12(self, x):                    # :32-40  12 coefficients x 8 u32 limbs
        self.append(bn.f12_to_u32(x))

    def append_g1(self, p):                      # :42-46
        self.append(bn.g1_to_u32(p))

    def append_g2(self, q):                      # :48-54  x.c0, x.c1, y.c0, y.c1
        self.append(bn.g2_to_u32(q))

    def get_challenge(self):                     # :56-65
        digest = hash_no_pad(self.state)
        u32 = []
        for x in digest:
            # BigUint::to_u32_digits drops high zero limbs (and returns [] for zero): the quirk of SURVEY a14
            if x == 0:
                continue
            u32.append(x & 0xFFFFFFFF)
            if x >> 32:
                u32.append(x >> 32)
        b = sum(d << (32 * i) for i, d in enumerate(u32))
        return b % bn.R


def inner_product(A, B):
    return bn.multi_pairing(A, B)


def sipp_prove_native(A, B):
    """returns (proof, trace) where proof is the reversed message list (src/prover_native.rs:78)"""
    assert len(A) == len(B)
    n = len(A)
    A, B = list(A), list(B)
    t = Transcript()
    proof = []
    for a, b in zip(A, B):
        t.append_g1(a)
        t.append_g2(b)
    Z = inner_product(A, B)
    proof.append(Z)
    t.append_fq12(Z)
    while n > 1:
        A1, A2 = A[:n // 2], A[n // 2:]
        B1, B2 = B[:n // 2], B[n // 2:]
        ZL = inner_product(A2, B1)
        ZR = inner_product(A1, B2)
        proof.append(ZL)
        t.append_fq12(ZL)
        proof.append(ZR)
        t.append_fq12(ZR)
        x = t.get_challenge()
        inv_x = pow(x, bn.R - 2, bn.R)
        A = [bn.g1_add(a1, bn.g1_mul(a2, x)) for a1, a2 in zip(A1, A2)]
        B = [bn.g2_add(b1, bn.g2_mul(b2, inv_x)) for b1, b2 in zip(B1, B2)]
        n //= 2
    proof.reverse()
    return proof


def sipp_verify_native(A, B, proof, check_final_pairing=True):
    """src/verifier_native.rs:14-85.  Returns (statement dict, obligations dict) -- the obligations are the
    three STARK input lists of src/verifier_circuit.rs:92-124 with their native outputs."""
    n = len(A)
    orig_A, orig_B = list(A), list(B)
    A, B = list(A), list(B)
    t = Transcript()
    proof = list(proof)
    for a, b in zip(A, B):
        t.append_g1(a)
        t.append_g2(b)
    orig_Z = proof.pop()
    Z = orig_Z
    t.append_fq12(Z)
    g1_ios, g2_ios, fq12_ios = [], [], []
    while n > 1:
        A1, A2 = A[:n // 2], A[n // 2:]
        B1, B2 = B[:n // 2], B[n // 2:]
        ZL = proof.pop()
        t.append_fq12(ZL)
        ZR = proof.pop()
        t.append_fq12(ZR)
        x = t.get_challenge()
        inv_x = pow(x, bn.R - 2, bn.R)
        newA, newB = [], []
        for a1, a2, b1, b2 in zip(A1, A2, B1, B2):
            na = bn.g1_add(a1, bn.g1_mul(a2, x))
            nb = bn.g2_add(b1, bn.g2_mul(b2, inv_x))
            g1_ios.append((a2, a1, x, na))           # {x: a2, offset: a1, exp_val: x}
            g2_ios.append((b2, b1, inv_x, nb))       # {x: b2, offset: b1, exp_val: inv_x}
            newA.append(na)
            newB.append(nb)
        z1 = bn.f12_mul(Z, bn.f12_pow(ZL, x))        # {x: Z_L, offset: Z, exp_val: x}
        fq12_ios.append((ZL, Z, x, z1))
        z2 = bn.f12_mul(z1, bn.f12_pow(ZR, inv_x))   # {x: Z_R, offset: Z Z_L^x, exp_val: inv_x}
        fq12_ios.append((ZR, z1, inv_x, z2))
        A, B, Z = newA, newB, z2
        n //= 2
    st = dict(A=orig_A, B=orig_B, Z=orig_Z, final_A=A[0], final_B=B[0], final_Z=Z)
    ok = True
    if check_final_pairing:
        ok = bn.pairing(A[0], B[0]) == Z
    return ok, st, dict(g1=g1_ios, g2=g2_ios, fq12=fq12_ios)


def statement_to_u32(st):
    """SIPPStatement flat layout (src/statements.rs:24-39): A (16n) | B (32n) | Z (96) | final_A | final_B | final_Z"""
    out = []
    for a in st["A"]:
        out += bn.g1_to_u32(a)
    for b in st["B"]:
        out += bn.g2_to_u32(b)
    out += bn.f12_to_u32(st["Z"]) + bn.g1_to_u32(st["final_A"]) + bn.g2_to_u32(st["final_B"]) + bn.f12_to_u32(st["final_Z"])
    return out


def exp_to_u32(e):
    return [(e >> (32 * i)) & 0xFFFFFFFF for i in range(8)]


def io_records(obl):
    """flat u32 arrays in the C-ABI record layout (x, offset, exp_val, output): 56 / 104 / 296 words per IO"""
    g1 = np.array([bn.g1_to_u32(x) + bn.g1_to_u32(o) + exp_to_u32(e) + bn.g1_to_u32(r) for x, o, e, r in obl["g1"]],
                  dtype=np.uint32).reshape(-1, 56)
    g2 = np.array([bn.g2_to_u32(x) + bn.g2_to_u32(o) + exp_to_u32(e) + bn.g2_to_u32(r) for x, o, e, r in obl["g2"]],
                  dtype=np.uint32).reshape(-1, 104)
    f12 = np.array([bn.f12_to_u32(x) + bn.f12_to_u32(o) + exp_to_u32(e) + bn.f12_to_u32(r) for x, o, e, r in obl["fq12"]],
                   dtype=np.uint32).reshape(-1, 296)
    return g1, g2, f12


def splitmix64(seed):
    s = seed & 0xFFFFFFFFFFFFFFFF
    while True:
        s = (s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        yield z ^ (z >> 31)


def synthetic_inputs(n, seed):
    """A_i = [s_i] G1, B_i = [t_i] G2 with s, t from SplitMix64 (replaces the reference's
    unseeded thread_rng at src/verifier_circuit.rs:202-204)."""
    g = splitmix64(seed)

    def scalar():
        v = 0
        for _ in range(4):
            v = (v << 64) | next(g)
        return v % bn.R or 1
    A = [bn.g1_mul(bn.G1, scalar()) for _ in range(n)]
    B = [bn.g2_mul(bn.G2, scalar()) for _ in range(n)]
    return A, B
