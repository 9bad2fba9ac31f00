"""What a verifier pays for the STARK proofs of one SIPP instance, counted from the proofs' shapes (no GPU, no circuit): proof
words, FRI query openings, Merkle path hashes and leaf permutations -- the price of level L-D: `world` ranks
prove 3 * world smaller proofs instead of 3, and the recursive verifier behind reference src/verifier_circuit.rs:133-147
(`verify_stark_proof_circuit`, one gadget per proof) grows with every one of them."""
import ctypes as C

from . import _lib

IO_WORDS = {0: 56, 1: 104, 2: 296, 3: 48, 4: 56, 5: 104}


def shape(kind, num_io):
    """(log_n, W, P, Q) of the STARK that proves `num_io` records of `kind` (sipp_stark_shape; pure host arithmetic)"""
    L = _lib.lib()
    v = [C.c_uint32() for _ in range(4)]
    rc = L.sipp_stark_shape(None, int(kind), int(num_io), *[C.byref(x) for x in v])
    if rc != 0:
        raise _lib.SippError(rc, "stark_shape")
    return tuple(x.value for x in v)


def fri_arities(log_n, rate_bits=1, cap_height=4, arity_bits=4, final_poly_bits=5):
    """FriReductionStrategy::ConstantArityBits as stark.hip fri_params_of reads it"""
    out, d = [], log_n
    while d > final_poly_bits and d + rate_bits - arity_bits >= cap_height and d >= arity_bits and len(out) < 32:
        out.append(arity_bits)
        d -= arity_bits
    return out, d


def stark_verifier_cost(kind, num_io, rate_bits=1, cap_height=4, arity_bits=4, final_poly_bits=5, num_queries=84):
    """one flat proof: its length in words and what verify_stark_proof does with it"""
    log_n, W, P, Q = shape(kind, num_io)
    log_m = log_n + rate_bits
    ar, final_bits = fri_arities(log_n, rate_bits, cap_height, arity_bits, final_poly_bits)
    cap = 4 << cap_height
    num_io_padded = 1 << (log_n - (3 if kind == 3 else 9))
    leaf_perms = lambda cols: (cols + 7) // 8 if cols > 4 else 0          # hash_or_noop
    words = 16 + 3 * cap + 2 * (2 * W + 2 * P + Q)
    words += len(ar) * cap + 2 * (1 << final_bits) + 1                     # commit caps, final polynomial (ext), pow witness
    per_query_words, per_query_perms, per_query_paths = 0, 0, 0
    for cols in (W, P, Q):
        sib = log_m - cap_height
        per_query_words += cols + 4 * sib
        per_query_perms += leaf_perms(cols)
        per_query_paths += sib
    lm = log_m
    for a in ar:
        lm -= a
        sib = max(0, lm - cap_height)
        per_query_words += 2 * (1 << a) + 4 * sib
        per_query_perms += leaf_perms(2 << a)
        per_query_paths += sib
    words += num_queries * per_query_words + num_io_padded * IO_WORDS[kind]
    # the challenger: statement (20 words), three caps, the opening set, per FRI layer a cap, the final polynomial, the witness
    observed = 20 + 3 * cap + 2 * (2 * W + 2 * P + Q) + len(ar) * cap + 2 * (1 << final_bits) + 1
    return {"kind": kind, "num_io": int(num_io), "log_n": log_n, "columns": [W, P, Q], "fri_rounds": len(ar), "proof_words": words,
            "query_openings": num_queries * (3 + len(ar)),              # Merkle openings: 3 initial trees + one per fold layer
            "merkle_path_hashes": num_queries * per_query_paths,          # two_to_one compressions up to the caps
            "leaf_permutations": num_queries * per_query_perms,          # hash_no_pad of the opened rows
            "challenger_permutations": (observed + 7) // 8,
            "constraint_evaluations": 1}                                   # the AIR at zeta: one per proof, W + P + Q openings wide


def instance_price(num_io, world=1, hardened=True, **cfg):
    """all proofs of one instance whose obligation lists (num_io = [g1, g2, fq12] records) are cut into `world` ranges (sipp_io_shard):
    totals over the 3 * world proofs (ranges without a record yield no proof)"""
    kinds = [4, 5, 2] if hardened else [0, 1, 2]
    tot = {"world": world, "proofs": 0, "proof_words": 0, "query_openings": 0, "merkle_path_hashes": 0, "leaf_permutations": 0,
           "challenger_permutations": 0, "opened_columns": 0}
    for rank in range(world):
        for k in range(3):
            first, count = _lib.io_shard(num_io[k], world, rank)
            if count == 0:
                continue
            c = stark_verifier_cost(kinds[k], count, **cfg)
            tot["proofs"] += 1
            for key in ("proof_words", "query_openings", "merkle_path_hashes", "leaf_permutations", "challenger_permutations"):
                tot[key] += c[key]
            tot["opened_columns"] += sum(c["columns"])
    tot["verifier_hashes"] = tot["merkle_path_hashes"] + tot["leaf_permutations"] + tot["challenger_permutations"]
    return tot
