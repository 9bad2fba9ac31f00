#!/usr/bin/env python3
"""bench.py -- SIPP STARK sub-prover benchmark (BASELINE.json metric: SIPP proof-gen wall-clock and
pairings-aggregated/s at n = 128).

A "step" is one pass of the hot path over one batch of synthetic input: the three starky sub-proofs
(G1 exp, G2 exp, Fq12 exp) behind reference src/verifier_circuit.rs:133-135 for ONE SIPP instance of
n = 128 pairings (127 / 127 / 14 IO records, seeded inputs, tests/golden/sipp_n128_ios.npz), from IO
records on the host to three flat proofs on the host, through the C ABI of include/sipp_hip.h.
The IO records (105 KB) are the only host input; everything of size O(N) is produced and consumed in HBM.

Multi-GPU (--gpus N, launched by torch.distributed.run): the path shards by independent SIPP instances
(level L-A): every rank proves its own instance, no data-path collective, weak
scaling; value = N * 128 / max-over-ranks step time.

AIR variant: since round 4 the timed region proves G1 / G2 with the HARDENED AIRs (API kinds 4 / 5, sipp_ctx_set_hardened) -- the plain
kinds' chord rule is forgeable for crafted statements; the plain kinds (this repository's reading of the
recalled upstream AIR) are timed beside it as `recalled_upstream_air`.  SIPP_BENCH_PLAIN_AIR=1 swaps the two.

A secondary object `io_sharded` times ONE larger instance cut into `world` ranges of obligations (level L-D, strong scaling,
no collective either).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters (spec)
NTT_KERNELS = ("lde_column", "lde_gather", "lde_mid", "ntt_dif_pass", "ntt_dit_pass", "bitrev_cols", "ntt_tree_gather", "ntt_tree_inv",
               "ntt_tree_mid", "ntt_tree_fwd")
PMC_FILE = "r06_pmc.json"   # rocprofv3 PMC passes of this same command (scripts/profile_round.sh), committed under profiles/


def load_ios(n):
    d = np.load(os.path.join(ROOT, "tests", "golden", "sipp_n%d_ios.npz" % n))
    return [d["g1"], d["g2"], d["fq12"]]


def cpu_baseline(ios, shapes, kinds, budget_s=150.0):
    """The CPU restatement (oracle/stark.c, OpenMP -- a `port`, not the reference's Rust binary, which cannot be built here)
    timed on the SAME workload as the GPU line: the three full-size n = 128 sub-proofs, one after the other, measured not
    scaled.  Threads = the box's CPU share (16 for one GPU; the oracle does not get faster beyond it: G1 with 32 IO records
    takes 7.8 s on 16 threads, 8.6 s on 64, 21 s on 256 -- scripts/omp_scaling.py).  If the first proof shows that the three
    would not fit `budget_s`, the other two are extrapolated by committed LDE cells and the object says so."""
    from tests import _oracle
    want = int(os.environ.get("SIPP_CPU_THREADS", "0")) or min(16, os.cpu_count() or 1)
    # omp_set_num_threads through the oracle: OMP_NUM_THREADS is read when the OpenMP runtime loads (with `import torch`, long
    # ago); `threads` is what omp_get_max_threads() reports afterwards, i.e. the team size the proofs below really run with
    threads = _oracle.set_num_threads(want)
    cells = [2.0 * (1 << s[0]) * (s[1] + s[2] + s[3]) for s in shapes]
    secs, measured = [], True
    for k in range(3):
        if k > 0 and secs[0] * sum(cells) / cells[0] > budget_s:
            measured = False
            secs.append(secs[0] * cells[k] / cells[0])
            continue
        t = time.time()
        pf = _oracle.stark_prove(kinds[k], ios[k])
        secs.append(time.time() - t)
        assert int(pf[1]) == kinds[k] and int(pf[2]) == shapes[k][0] and int(pf[4]) == shapes[k][1]      # the same AIR variant / shape as the GPU line
    total = sum(secs)
    n = ios[0].shape[0] + 1
    return {"value": n / total, "unit": "pairings/s", "cores": threads, "kind": "port",
            "seconds": [round(x, 2) for x in secs], "measured": measured,
            "sample": "oracle/stark.c (OpenMP, %d threads) on the full workload of this line: the G1 / G2 / Fq12 sub-proofs (kinds %s) of the "
                      "n = %d instance took %s s%s" % (threads, "/".join(str(k) for k in kinds), n, " + ".join("%.1f" % x for x in secs),
                                                     "" if measured else " (only the first measured, the others scaled by committed cells)")}


def verify_proofs(proofs, lib_ms=None):
    """every proof this run produced goes, OUTSIDE every timed region, through BOTH verifiers: the library's own (sipp_stark_verify:
    host code of the product, what a user of the library runs -- the generators' verify_stark_proof / data.verify of the reference) and
    the CPU checker (oracle/stark.c's verifier, test infrastructure): a bench line for proofs nobody checked is a claim, not a result.
    Returns True when every non-empty proof is accepted by both; lib_ms (a list) receives the library verifier's time per proof."""
    try:
        import sipp_amd
        from tests import _oracle
        ok = True
        for pf in proofs:
            if len(pf):
                pf = np.ascontiguousarray(pf)
                t = time.perf_counter()
                lib_ok = sipp_amd.stark_verify(pf) == 0
                if lib_ms is not None:
                    lib_ms.append(round(1e3 * (time.perf_counter() - t), 2))
                ok = ok and lib_ok and _oracle.stark_verify(pf) == 0
        return bool(ok)
    except Exception as e:      # a checker that cannot run (build failure, missing compiler) is `verified: false`, never a rank that
        VERIFY_ERRORS.append("%s: %s" % (type(e).__name__, e))   # leaves the collective the other ranks are waiting in
        return False


VERIFY_ERRORS = []


def prepare_checker(rank, barrier):
    """the oracle library is built ONCE, by rank 0, BEFORE the timed region (liboracle.so is git-ignored: on a fresh tree all N
    ranks would otherwise run the same `make` at once, right after the timing); the others load it after the barrier.  A failure
    is recorded, not raised: every rank must reach the reductions that follow."""
    def build():
        try:
            from tests import _oracle
            _oracle.load()
        except Exception as e:
            VERIFY_ERRORS.append("%s: %s" % (type(e).__name__, e))
    if rank == 0:
        build()
    barrier()
    if rank != 0:
        build()


def all_ranks_true(flag, device):
    """a per-rank boolean folded over the ranks (min): the line says `verified` only if every rank's proofs were accepted"""
    from sipp_amd import dist_util
    lo, _ = dist_util.min_max_over_ranks(1.0 if flag else 0.0, device=device)
    return bool(lo > 0.5)


def projected_shards(n, hardened, world=8):
    path = os.path.join(ROOT, "profiles", "r06_io_shard_projection_%s_n%d_%d.json" % ("hardened" if hardened else "plain", n, world))
    if not os.path.exists(path):
        return None
    d = json.load(open(path))
    return {"projection": True, "world": d["world"], "whole_instance_on_one_gpu_ms": d["whole_instance_on_one_gpu_ms"],
            "slowest_shard_ms": d["slowest_shard_ms"], "projected_speedup": d["projected_speedup"],
            "arena_bytes_per_rank_max": max(s["arena_bytes"] for s in d["shards_one_at_a_time"]), "source": os.path.basename(path)}


def price_vs_world1(lists, world, hardened):
    from sipp_amd import proof_cost
    one = proof_cost.instance_price(lists, 1, hardened)
    w = proof_cost.instance_price(lists, world, hardened)
    return {"proofs": w["proofs"], "proof_words": w["proof_words"], "query_openings": w["query_openings"], "verifier_hashes": w["verifier_hashes"],
            "proof_words_vs_world1": w["proof_words"] / one["proof_words"], "verifier_hashes_vs_world1": w["verifier_hashes"] / one["verifier_hashes"]}


def outer_plonk_leg(device=0, log_n=18, steps=5, verify=True):
    """Secondary leg, outside the timed region of `value`: plonky2's outer prove() at the SHAPE of the
    reference's circuit configuration -- CircuitConfig::standard_ecc_config (reference src/verifier_circuit.rs:213: 136 wires, 80 routed,
    2 challenges, quotient degree factor 8, FRI rate_bits 3 / cap 4 / 28 queries / 16 grinding bits / arity 16) -- on a SYNTHETIC circuit
    of 2^log_n rows (tools/plonk_synth.py, round 6: RECURSION-SHAPED -- Poseidon (the whole permutation, 118 constraints of degree 7),
    U32 multiply-add with 2-bit limb range checks, random access, reducing, arithmetic, base-sum, constant and public-input gates in
    three selector groups; the reference's own gate set and witness live in un-vendored crates) through sipp_plonk_prove_gates: wires
    commitment, Z / partial products, the gate constraints interpreted inside the quotient kernel, quotient commitment, openings, FRI.
    constants_sigmas is committed once, outside the timing, as plonky2 does at circuit-build time.  A timed step is the WHOLE prove():
    witness generation on the device (sipp_plonk_generate_witness: the gates' generators, one lane per row, in place on the wire table
    -- until round 6's last step a numpy generator on one host core, 2.3 s beside a 72 ms proof; the circuit is CHAINED: Poseidon hash
    chains of 64 links feeding arithmetic and reducing rows through copy constraints, so the generators run level by level as a build-time
    schedule, two launches per level replayed from a hipGraph: sipp_plonk_generate_witness_levels) and then everything below it.  What
    stays on the host is what plonky2 does outside prove(): building the circuit and assigning the input cells (`host_circuit_and_inputs_s`).
    The C port of the generators (oracle/plonk_witness.c, OpenMP) runs once on the same inputs: its table must equal the device's
    (`witness_matches_cpu_port`) and its time is the CPU column of this step."""
    import ctypes as C
    import torch
    import sipp_amd
    from sipp_amd._lib import to_device
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import plonk_synth as ps
    n = 1 << log_n
    W, R, D, CH, rate_bits, cap_h, nq, pow_bits = 136, 80, 8, 2, 3, 4, 28, 16
    circ = ps.circuit(W, R)
    K = circ["num_constants"]
    L = sipp_amd.lib()
    cols_timed = W + CH * ((R + D - 1) // D) + CH * D                  # wires, zs_partial_products, quotient chunks
    ws = int(8 * n * (9 * (K + R + cols_timed) + 64) + (4 << 30))
    ctx = sipp_amd.Ctx(device=device, workspace_bytes=ws)
    try:
        pis = [3, 1, 4, 1, 5, 9, 2, 6]
        st = torch.zeros((1, 12), dtype=torch.int64, device="cuda")
        st[0, :len(pis)] = torch.tensor(pis, dtype=torch.int64)
        pih = [int(x) & 0xFFFFFFFFFFFFFFFF for x in ctx.poseidon_permute(st)[0, :4].tolist()]    # hash_n_to_hash_no_pad of <= 8 inputs
        CHAIN = 64                                                                      # links of a Poseidon hash chain (Merkle-path-like)
        t0 = time.perf_counter()
        wires, cs, _gate = ps.witness(circ, log_n, 2026, pih, inputs_only=True, chain_len=CHAIN)   # input cells, constants, sigmas: the caller's side
        sc = ps.chain_schedule(log_n, CHAIN)                                          # circuit-build time: the generators' levels
        t_inputs = time.perf_counter() - t0
        gens = ps.generators(circ)
        d_w, d_cs = to_device(wires), to_device(cs)
        d_k = d_cs[:K]
        sched = sipp_amd.PlonkSchedule.from_dict(sc)
        gp = sipp_amd.PlonkParams(R, D, CH)
        fp = sipp_amd.FriParams()
        fp.rate_bits, fp.cap_height, fp.pow_bits, fp.num_queries, fp.pow_rule, fp.hiding = rate_bits, cap_h, pow_bits, nq, 0, 0
        L.sipp_fri_const_arity(C.byref(fp), 4, 5, log_n)
        gc = sipp_amd.PlonkCircuit.from_dict(circ)
        digest = (0x53495050, 0x6f757465, 0x72706c6f, 0x6e6b3035)
        cs_or, cs_cap, keep = ctx.commit_ex(d_cs, log_n, rate_bits, cap_h)          # once per circuit

        def generate():                                                              # in place; input cells are never written: repeatable
            ctx.plonk_generate_witness_levels(d_w, d_k, log_n, gens, pih, sched)

        def prove():
            generate()
            return ctx.plonk_prove_gates(d_w, d_cs, log_n, gp, fp, gc, digest, pis, cs_oracle=cs_or)
        prove()
        # witness generation alone: the captured hipGraph against the same launches one by one (the call ends with a stream sync)
        wit = {}
        for name, route in (("graph_replay_ms", 0), ("launch_by_launch_ms", 4)):
            L.sipp_ctx_set_kernel_routes(ctx.h, route)
            generate()
            t0 = time.perf_counter()
            for _ in range(4 * steps):
                generate()
            wit[name] = 1e3 * (time.perf_counter() - t0) / (4 * steps)
        L.sipp_ctx_set_kernel_routes(ctx.h, 0)
        wit_ms = wit["graph_replay_ms"]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):                                                        # the timed steps: the whole prove(), profile off (graph route)
            pf = prove()
        ctx.sync()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        ctx.profile(True)                                                             # a second pass for the per-kernel breakdown
        ctx.profile_reset()
        for _ in range(steps):
            prove()
        ctx.sync()
        rep = {k: v["ms"] / steps for k, v in ctx.profile_report().items()}
        ctx.profile(False)
        del keep
        # the same prove() through the HOST-pointer boundary (sipp_circuit_build once, then sipp_circuit_prove: the wire table with its input
        # cells crosses PCIe on every proof -- 8 N num_wires bytes -- and the proof comes back to host memory): the PCIe-inclusive figure
        host_to_host = None
        try:
            ctx.sync()
            data = sipp_amd.CircuitData(ctx, log_n, gp, fp, gc, cs, gens, sched=sc, digest=digest)
            try:
                pf_h = data.prove(wires, pis)
                t0 = time.perf_counter()
                for _ in range(steps):
                    pf_h = data.prove(wires, pis)
                host_to_host = {"ms_per_proof": 1e3 * (time.perf_counter() - t0) / steps, "h2d_bytes_per_proof": int(wires.nbytes),
                                "entry_points": "sipp_circuit_build, sipp_circuit_prove", "same_proof": bool(len(pf_h) == len(pf) and (pf_h == pf).all()),
                                "verified": data.verify(pf_h) == (0, 0)}
            finally:
                data.close()
        except Exception as e:                                                        # noqa: BLE001 -- a secondary figure
            host_to_host = {"error": repr(e)}
        wit.update({"levels": int(sc["n_levels"]), "launches": int(np.count_nonzero(np.diff(sc["level_offsets"])) + np.count_nonzero(np.diff(sc["copy_offsets"]))),
                    "copies": int(len(sc["copy_src"])), "chain_len": CHAIN})
        ntt_names = [k for k in rep if k.startswith(("ntt_", "lde_", "bitrev"))]
        leaf_names = [k for k in rep if k.startswith("poseidon_leaves")]
        ntt_ms, leaf_ms = sum(rep[k] for k in ntt_names), sum(rep[k] for k in leaf_names)
        # algorithmic bytes of the transforms: values -> coefficients -> blowup-8 LDE per column of wires and zs (8 N read, 8 N + 64 N
        # written), coefficients -> LDE for the quotient chunks (8 N + 64 N), the quotient's own coset iNTT (2 x 8 N x 16 B)
        zs_cols, q_cols = CH * ((R + D - 1) // D), CH * D
        ntt_bytes = 80.0 * n * (W + zs_cols) + 72.0 * n * q_cols + 16.0 * 8 * n * CH
        m = 8 * n
        perms = m * sum((c + 7) // 8 for c in (W, zs_cols, q_cols))
        leaf_bytes = 8.0 * m * (W + zs_cols + q_cols) + 3 * 32.0 * m
        verified = None
        cpu_wit = {}
        if verify:
            from tests import _oracle
            from sipp_amd._lib import to_host
            t_c = time.perf_counter()
            ref_w = _oracle.plonk_generate_witness_levels(wires, cs[:K], log_n, gens, pih, sc)
            cpu_wit = {"cpu_witness_generation_s": time.perf_counter() - t_c, "cpu_witness_kind": "port (oracle/plonk_witness.c, one thread, incl. one copy of the table)",
                       "cpu_cores": 1, "witness_matches_cpu_port": bool((to_host(d_w) == ref_w).all())}
            del ref_w
            ofp = _oracle.fri_params(rate_bits=rate_bits, cap_height=cap_h, pow_bits=pow_bits, num_queries=nq, pow_rule=0, hiding=0, arity_bits=4,
                                     final_poly_bits=5, degree_bits=log_n)
            # both verifiers: the library's own (sipp_plonk_verify_gates: data.verify's part for the outer proof) and the oracle's
            from tests import _verify
            t_v = time.perf_counter()
            lib_ok = _verify.lib_plonk_verify(pf, cs_cap, _oracle.plonk_params(R, D, CH), ofp, circ, digest) == 0
            lib_verify_ms = round(1e3 * (time.perf_counter() - t_v), 2)
            verified = lib_ok and _oracle.plonk_verify_gates(pf, cs_cap, _oracle.plonk_params(R, D, CH), ofp, circ, digest) == 0 and cpu_wit["witness_matches_cpu_port"]
        return {"what": "plonky2 prove() INCLUDING witness generation (sipp_plonk_generate_witness_levels, then sipp_plonk_prove_gates) on a synthetic chained circuit with gates, generators and the level schedule as data",
                "shape": {"degree_bits": log_n, "num_wires": W, "num_routed_wires": R, "num_constants": K, "num_challenges": CH, "quotient_degree_factor": D,
                          "rate_bits": rate_bits, "cap_height": cap_h, "num_queries": nq, "pow_bits": pow_bits, "arity": 16,
                          "gates": circ["gate_names"], "num_gate_constraints": circ["num_gate_constraints"], "program_words": int(len(circ["programs"])),
                          "constraints_per_gate": [int(g[5]) for g in circ["gates"]]},
                "config_ref": "CircuitConfig::standard_ecc_config (reference src/verifier_circuit.rs:213); degree of the reference's circuit unknown "
                              "(built by un-vendored crates): 2^%d rows here" % log_n,
                "ms_per_proof": ms, "steps": steps, "proof_words": int(len(pf)), "verified": verified,
                **({"library_verifier_ms": lib_verify_ms} if verify else {}),
                # HEADLINE, next to ms_per_proof (which contains it): witness generation runs on the device, level by level (values that travel
                # between rows through copy constraints follow the build-time schedule).  The host keeps what plonky2 does outside prove()
                "witness_generation_s": wit_ms * 1e-3, "witness_generation_ms": wit_ms, "prove_below_witness_ms": ms - wit_ms, "witness_generation": wit,
                "end_to_end_s_per_proof": ms * 1e-3, "host_circuit_and_inputs_s": t_inputs, "host_to_host": host_to_host, **cpu_wit,
                "kernel_ms_per_proof": {k: round(v, 3) for k, v in sorted(rep.items(), key=lambda kv: -kv[1])},
                "roofline": {"transforms": {"bound": "hbm", "kernels": sorted(ntt_names), "algorithmic_bytes": ntt_bytes, "ms": ntt_ms,
                                            "achieved": ntt_bytes / (ntt_ms * 1e-3) / 1e9 if ntt_ms else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                            "frac": ntt_bytes / (ntt_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ntt_ms else None},
                             "leaf_hashing": {"bound": "valu", "kernels": sorted(leaf_names), "permutations": perms, "ms": leaf_ms,
                                              "perms_per_s": perms / (leaf_ms * 1e-3) if leaf_ms else None,
                                              "algorithmic_bytes": leaf_bytes, "achieved": leaf_bytes / (leaf_ms * 1e-3) / 1e9 if leaf_ms else None,
                                              "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": leaf_bytes / (leaf_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if leaf_ms else None},
                             "quotient_with_gates_ms": rep.get("plonk_quotient"),
                             "plonk_quotient": plonk_quotient_roofline(circ, n, W, K, R, CH, D, zs_cols, rep.get("plonk_quotient"))}}
    finally:
        ctx.close()


def plonk_quotient_roofline(circ, n, W, K, R, CH, D, zs_cols, ms):
    """the outer prover's quotient kernel (permutation terms + every gate's program at every point of the blowup-8 coset, one lane per
    point) against both bounds: HBM (each LDE cell of constants_sigmas, wires, Z / partial products read once, the quotient values
    written) and VALU (field products the programs ask for: per monomial its factors and a non-unit coefficient, per constraint the
    filter and the alpha fold of every challenge; ~26 wave instructions per product at one instruction per 4 cycles)"""
    if not ms:
        return None
    # what the kernel runs since the last step of round 6 (plonk.hip compile_gates): per gate its DISTINCT monomials, each evaluated once
    # (a pure power w^k by squaring, or ONE product when w^(k-1) came just before; a mixed monomial its factors) and weighted by one
    # alpha-folded coefficient per challenge; `as_written` = the products the programs ask for constraint by constraint (rounds 5 - 6)
    prog = [int(x) for x in circ["programs"]]
    products = as_written = monomials = distinct = 0
    for (_si, _row, lo, hi, off, nc) in circ["gates"]:
        w = off
        products += (hi - lo) + CH                           # the gate's filter, filter x folded sum per challenge
        as_written += (hi - lo)
        seen = set()
        for _ in range(nc):
            nm = prog[w]
            w += 1
            for _m in range(nm):
                coef, nf = prog[w], prog[w + 1]
                seen.add(tuple(sorted((prog[w + 2 + 2 * i], prog[w + 3 + 2 * i]) for i in range(nf))))
                w += 2 + 2 * nf
                as_written += max(0, nf - 1) + (1 if abs(coef) != 1 and nf else 0)
                monomials += 1
            as_written += 1 + CH                             # filter x constraint, one alpha fold per challenge
        distinct += len(seen)
        pure = sorted((f[0], len(f)) for f in seen if f and f[0] == f[-1])
        for i, (op, k) in enumerate(pure):
            chained = i > 0 and pure[i - 1] == (op, k - 1)
            products += 1 if chained and k > 1 else (k.bit_length() - 1) + bin(k).count("1") - 1
        products += sum(len(f) - 1 for f in seen if f and f[0] != f[-1])
        products += CH * len(seen)
    m = 8 * n
    bytes_alg = 8.0 * m * (K + R + W + zs_cols) + 8.0 * m * CH
    peak_products = 1024 * 64 * 2.2e9 / (4 * 26)             # lanes x clock / (cycles per instruction x instructions per product)
    chunks = (R + D - 1) // D
    perm_products = CH * (4 * R + 2 * chunks + 2)            # the permutation terms beside the gates: per routed wire beta_k x, the numerator,
    ach = (products + perm_products) * m / (ms * 1e-3)       # beta sigma and the denominator; per chunk the two sides of check_partial_products
    return {"bound": "valu", "ms": ms, "gate_products_per_point": products, "gate_products_per_point_as_written": as_written,
            "monomials": monomials, "distinct_monomials": distinct, "permutation_products_per_point": perm_products, "points": m,
            "achieved_products_per_s": ach,
            "peak_products_per_s_est": peak_products, "frac_valu_est": ach / peak_products,
            "hbm": {"algorithmic_bytes": bytes_alg, "achieved": bytes_alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": bytes_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}}


def air_revision():
    """the AIR is this repository's own specification (tools/air_gen.py): identify the revision the numbers belong to"""
    import hashlib
    h = hashlib.sha256(open(os.path.join(ROOT, "data", "air_tables.h"), "rb").read()).hexdigest()[:12]
    return "data/air_tables.h sha256 %s (permuted lookups with independent (beta, gamma), statement-bound Fiat-Shamir)" % h


def ntt_roofline(shapes, kernel_ms_serial, pmc=None):
    """second roofline object, for the NTT / LDE kernels (the HBM-class kernels north_star names), from the SERIAL kernel
    times of this run.  Algorithmic bytes: SURVEY section 8(d) -- per committed column 8 N read + 8 N coefficients +
    16 N LDE written (iNTT 2 + LDE 3 minus the intermediate the fused kernels never write = 32 N B per column of W and P; the
    Q chunk columns and the quotient iNTT are two orders of magnitude smaller and left out)."""
    if not kernel_ms_serial:
        return None
    names = NTT_KERNELS
    ms = sum(kernel_ms_serial.get(k, 0.0) for k in names)
    alg = sum(32.0 * (1 << s[0]) * (s[1] + s[2]) for s in shapes)
    if ms <= 0:
        return None
    ach = alg / (ms * 1e-3) / 1e9
    out = {"bound": "hbm", "kernels": list(names), "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
           "algorithmic_bytes_per_step": alg, "serial_ms_per_step": ms,
           "note": "VALU-bound in practice: ~20 Goldilocks products (26 instructions each) + 39 modular add/sub per 32 bytes; "
                   "see DESIGN.md section 4"}
    # every sweep's own bound, from the counter passes of this same command (profiles/PMC_FILE: each kernel ALONE
    # on the chip): HBM bytes it really moves per launch (2 x FETCH_SIZE + WRITE_SIZE) over its launch time = achieved TB/s against the
    # 8 TB/s peak, and the share of its SIMD cycles the VALUs were busy -- a sweep near 0.8 busy at 3 - 4 TB/s gains nothing from moving
    # fewer bytes; only the strided inverse sweep of the long columns is memory-bound (profiles/r06_ab_transforms.txt)
    if pmc and pmc.get("ntt") and pmc.get("transforms_alone"):
        per = []
        for k, al in pmc["transforms_alone"].items():
            tr = pmc["ntt"]["kernels"].get(k)
            if not tr or not tr.get("launches") or not al.get("avg_launch_ms"):
                continue
            b = (2.0 * tr["FETCH_SIZE_kb_sum"] + tr["WRITE_SIZE_kb_sum"]) * 1024.0 / tr["launches"]
            per.append({"kernel": k, "ms_per_launch_alone": al["avg_launch_ms"], "traffic_bytes_per_launch": b,
                        "achieved_tbps": b / (al["avg_launch_ms"] * 1e-3) / 1e12, "frac_of_hbm_peak": b / (al["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "valu_busy_frac": al.get("valu_busy_frac"), "cycles_per_inst": al.get("cycles_per_inst")})
        out["per_kernel"] = sorted(per, key=lambda e: -e["ms_per_launch_alone"])
        if pmc["ntt"].get("traffic_bytes_per_instance"):
            out["traffic_bytes_per_step"] = pmc["ntt"]["traffic_bytes_per_instance"]
            out["traffic_over_algorithmic"] = pmc["ntt"]["traffic_bytes_per_instance"] / alg
    return out


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port <free> bench.py <argv>` as a CHILD process (never exec: this process may not be
    replaced) and pass its output and exit code through.  Called before anything in this process touches the GPU."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:          # rank 0's JSON line (and anything else the ranks print) as it comes
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=128, help="pairings per SIPP instance (fixture must exist)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--inflight", type=int, default=None,
                    help="instances proved concurrently for the secondary `pipelined` figure (1 = skip); default 5 on one "
                         "rank, 1 (skipped) when launched with more: 8 ranks x 5 slots x 3 worker threads would put 120 host "
                         "threads on one node for a secondary figure")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # invoked bare (`python bench.py --gpus N`): start the N ranks ourselves, exactly as the driver's launcher would.  Nothing
        # in this process has touched the GPU (torch is not even imported yet), the ranks are fresh children, rank 0's one JSON
        # line is relayed and the launcher's exit code is ours.
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.inflight is None:
        args.inflight = 5 if int(os.environ.get("WORLD_SIZE", "1")) == 1 else 1

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        # a launcher that supplies the wrong WORLD_SIZE: a one-rank number must not be reported as the N-GPU leg
        raise SystemExit("bench.py --gpus %d needs %d ranks but the launcher set WORLD_SIZE=%d: launch it as `python -m "
                         "torch.distributed.run --nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 --master-port P bench.py "
                         "--gpus %d ...`, or bare (`python bench.py --gpus %d` starts its own ranks)"
                         % (args.gpus, args.gpus, world, args.gpus, args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the SIPP HIP path has no CPU fallback")
    # SIPP_BENCH_REHEARSAL=1: every rank on GPU 0 over gloo -- only to rehearse the N > 1 code path on a one-GPU box
    # (the driver's scaling runs use one GPU per rank over RCCL, the default)
    rehearsal = bool(int(os.environ.get("SIPP_BENCH_REHEARSAL", "0")))
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    from sipp_amd import dist_util
    # SIPP_BENCH_SINGLE_RANK_GROUP=1: a process group of ONE rank, so that a one-GPU box runs the RCCL branch of the timing contract
    red_device = dist_util.init_process_group(world, local_rank, rehearsal,   # "cuda" over RCCL, "cpu" over gloo / alone
                                              single_rank_group=bool(int(os.environ.get("SIPP_BENCH_SINGLE_RANK_GROUP", "0"))))

    import sipp_amd
    ios = load_ios(args.n)
    # one ctx (= one HIP stream + workspace arena) and one host thread per STARK: the three sub-proofs are
    # independent, so their thin kernels and Fiat-Shamir round trips overlap.
    # The threads are the library's own (sipp_instance_prove = 3 x sipp_prove_async + sipp_wait).
    # stream levels of G1, G2, Fq12; None = sipp_amd.Instance's default for the AIR variant
    prios = os.environ["SIPP_BENCH_PRIOS"].split(",") if "SIPP_BENCH_PRIOS" in os.environ else None
    hardened = not int(os.environ.get("SIPP_BENCH_PLAIN_AIR", "0"))
    kinds = [4, 5, 2] if hardened else [0, 1, 2]
    ws = [sipp_amd.lib().sipp_workspace_bytes(kinds[k], ios[k].shape[0]) for k in range(3)]
    inst = sipp_amd.Instance([a.shape[0] for a in ios], devices=(local_rank,) * 3, priorities=prios, hardened=hardened)
    ctxs = inst.distinct_ctxs()
    if inst.single_ctx:
        ctxs = ctxs * 3
    ctx = ctxs[0]
    shapes = [ctx.shape(k, ios[k].shape[0]) for k in range(3)]
    serial = bool(int(os.environ.get("SIPP_BENCH_SERIAL", "0")))

    proof_ms = [0.0, 0.0, 0.0]

    def step():
        if serial:
            res = []
            for k in range(3):
                t = time.perf_counter()
                res.append(ctxs[k].prove(k, ios[k]))
                proof_ms[k] += 1e3 * (time.perf_counter() - t)
            return res
        return inst.prove(ios)

    def device_sync():
        torch.cuda.synchronize()
        inst.sync()

    def barrier():
        dist_util.barrier(device_sync)

    def start_profiling():
        for c in inst.distinct_ctxs():
            c.profile(True)
            c.profile_reset()
        proof_ms[:] = [0.0, 0.0, 0.0]

    prepare_checker(rank, barrier)
    # W untimed warm-up steps, then exactly K steps between barrier + synchronize, max over ranks
    elapsed, proofs = dist_util.timed_steps(step, args.steps, args.warmup, sync=device_sync, device=red_device,
                                            before_timing=start_profiling)
    proofs = [p.copy() for p in proofs]           # the instance's output buffers are reused by the later legs
    # every rank pushes the proofs of its LAST timed step through the oracle's verifier (outside the timed region)
    lib_verify_ms = []
    verified_main = all_ranks_true(verify_proofs(proofs, lib_verify_ms), red_device)
    prof = {}
    for c in inst.distinct_ctxs():
        for k, v in c.profile_report().items():
            e = prof.setdefault(k, {"calls": 0, "ms": 0.0})
            e["calls"] += v["calls"]
            e["ms"] += v["ms"]
        c.profile(False)

    def dist_util_max(x):
        return dist_util.max_over_ranks(x, device=red_device)

    # secondary figure, outside the timed region: `inflight` independent instances proved concurrently on this GPU
    # (3 streams each).  One instance leaves issue slots idle in its latency-bound phases (chains, Fiat-Shamir round
    # trips, FRI tail); a server with a queue of proofs fills them.  `value` above stays the single-instance number.
    pipelined = None
    if args.inflight > 1 and not serial and sum(ws) * (args.inflight + 1) < (240 << 30):
        # sipp_instances_prove: `inflight` slots of three ctxs take the instances of a queue from a shared counter
        queue = sipp_amd.InstanceQueue([a.shape[0] for a in ios], in_flight=args.inflight, device=local_rank, priorities=prios,
                                       hardened=hardened)
        queue.prove([ios] * args.inflight)
        barrier()
        tp = time.perf_counter()
        qproofs = queue.prove([ios] * (args.inflight * args.steps))
        barrier()
        dtp = dist_util_max(time.perf_counter() - tp)
        # the last instance of the queue through the verifier; every instance must equal the single-instance proofs word for word
        q_ok = verify_proofs(qproofs[-1]) and all(len(a) == len(b) and (a == b).all() for inst_p in qproofs for a, b in zip(inst_p, proofs))
        pipelined = {"instances_in_flight": args.inflight, "ms_per_instance": 1e3 * dtp / (args.steps * args.inflight),
                     "value": args.n * world * args.inflight * args.steps / dtp, "unit": "pairings/s",
                     "verified": all_ranks_true(q_ok, red_device),
                     "entry_point": "sipp_instances_prove (queue of %d instances)" % (args.inflight * args.steps)}
        queue.close()

    # one extra step with the three proofs run one after the other (outside the timed region): per-kernel times without the
    # other proofs' kernels competing for the SIMDs
    kernel_ms_serial = None
    if rank == 0 and not serial:
        for c in inst.distinct_ctxs():
            c.profile(True)
            c.profile_reset()
        for k in range(3):
            ctxs[k].prove(k, ios[k])
        kernel_ms_serial = {}
        for c in inst.distinct_ctxs():
            for k, v in c.profile_report().items():
                kernel_ms_serial[k] = kernel_ms_serial.get(k, 0.0) + v["ms"]
            c.profile(False)

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = dist_util.whole_job_rate(args.n, world, elapsed / args.steps)
        # dominant kernel: Poseidon leaf hashing, ONE STATE PER LANE (poseidon_leaves_kernel: every tree of more than 2^16 leaves).
        # Algorithmic bytes per launch: every LDE cell of the batch is read once (8 B), one 32-B digest per leaf is written.  The
        # thin trees' two-lane kernel (the Fq12 STARK) is a different kernel and has its own entry; batches of <= 4 columns are
        # not hashed at all (hash_or_noop: a copy) and are in neither.
        leaf = {"one": {"bytes": 0.0, "perms": 0.0, "batches": 0}, "pair": {"bytes": 0.0, "perms": 0.0, "batches": 0}}
        for (log_n, W, P, Q) in shapes:
            m = 2 << log_n
            for cols in (W, P, Q):
                if cols <= 4:
                    continue
                e = leaf["one" if m > 65536 else "pair"]
                e["bytes"] += 8.0 * m * cols + 32.0 * m
                e["perms"] += m * ((cols + 7) // 8)
                e["batches"] += 1
        leaf_perms = leaf["one"]["perms"] + leaf["pair"]["perms"]
        pmc, pmc_state = {}, "no counters for this size (the PMC passes profile the n = 128 line)"
        tpath = os.path.join(ROOT, "profiles", PMC_FILE)
        if args.n == 128 and os.path.exists(tpath):
            from sipp_amd import build as sipp_build
            pmc = json.load(open(tpath))
            pmc_state = "current"
            if pmc.get("kinds", [0, 1, 2]) != kinds:      # counters of the other AIR variant do not belong to this line
                pmc, pmc_state = {}, "counters of the other AIR variant: not used"
            elif pmc.get("source_sha256") != sipp_build.source_hash():
                # the kernels changed after the PMC passes were taken (scripts/profile_round.sh): stale counters are refused, not reported
                pmc, pmc_state = {}, ("STALE: profiles/%s was collected on other kernel sources (sha256 %s...); re-run "
                                      "scripts/profile_round.sh" % (PMC_FILE, str(pmc.get("source_sha256"))[:12]))
        # nominal issue peak: 1024 SIMDs, one wave64 instruction per 2 cycles (a PAIR of plain 32-bit forms per quad-cycle) at 2.4 GHz
        issue_peak = 256 * 4 * 2.4e9 / 2 / 1e9    # G wave-instructions/s

        def leaf_entry(kind, prof_name):
            lk_ = prof.get(prof_name, {"calls": 0, "ms": 0.0})
            n_l = lk_["calls"]
            if not n_l or lk_["ms"] <= 0:
                return None, lk_
            avg = lk_["ms"] / n_l
            bpl = leaf[kind]["bytes"] * args.steps / n_l
            ach = bpl / (avg * 1e-3) / 1e9
            st = pmc.get("leaf_" + kind, {})
            al = pmc.get("leaf_" + kind + "_alone") or {}
            vi = st.get("valu_insts_per_launch")
            valu_obj = None
            if vi:
                ach_v = vi / (avg * 1e-3) / 1e9
                valu_obj = {"unit": "G wave-instructions/s", "achieved": ach_v, "peak_nominal": issue_peak, "frac_nominal": ach_v / issue_peak,
                            "insts_per_launch": vi}
                if al.get("eff_clock_ghz") and al.get("cycles_per_inst"):
                    # calibrated ceiling of THIS kernel's instruction mix at the clock the chip holds under it (both from the PMC pass
                    # of this command, profiles/PMC_FILE; method and the counters' meaning: profiles/README.md, r05_enc_rates.txt):
                    #   a VALU instruction occupies its SIMD for 4 cycles unless it issues as one of a pair (plain 32-bit VOP1 / VOP2
                    #   forms only); cycles_per_inst = 4 (SQ_ACTIVE_INST_VALU - SQ_ACTIVE_INST_VALU2) / SQ_INSTS_VALU
                    clk, cpi = al["eff_clock_ghz"], al["cycles_per_inst"]
                    cal = 1024.0 * clk / cpi
                    valu_obj.update({"eff_clock_ghz": clk, "cycles_per_inst": cpi,
                                     "mix_cost": cpi / 2.0,      # issue slots of the nominal peak (2 cycles) per instruction
                                     "peak_calibrated": cal, "frac_of_calibrated": ach_v / cal,
                                     # the kernel ALONE on the chip, as the counter pass saw it (static): launch time and the share of
                                     # its SIMD cycles the VALUs were busy (4 x SQ_ACTIVE_INST_VALU / (1024 x GRBM_GUI_ACTIVE / 8))
                                     "alone": {"avg_launch_ms": al["avg_launch_ms"], "achieved": vi / (al["avg_launch_ms"] * 1e-3) / 1e9,
                                               "frac_of_calibrated": vi / (al["avg_launch_ms"] * 1e-3) / 1e9 / cal,
                                               "valu_busy_frac": al.get("valu_busy_frac")}})
            return {"kernel": prof_name, "achieved": ach, "frac": ach / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": bpl,
                    "launches": n_l, "avg_launch_ms": avg, "traffic": st.get("traffic_bytes_per_launch"),
                    "perms_per_s": leaf[kind]["perms"] * args.steps / (lk_["ms"] * 1e-3),
                    # the kernel's real bound: wave-level VALU instructions per launch (PMC SQ_INSTS_VALU of this same command,
                    # static) over the live launch time, against the issue peak (MI355X_MICROARCH.md, wave scheduling)
                    "valu": valu_obj}, lk_

        one, lk = leaf_entry("one", "poseidon_leaves")
        pair, lkp = leaf_entry("pair", "poseidon_leaves_pair")
        if one is None:          # a size without fat trees (n < 128): the thin kernel is the dominant one
            one, lk = pair, lkp
        achieved, launches, avg_ms = (one["achieved"], one["launches"], one["avg_launch_ms"]) if one else (0.0, 0, 0.0)
        bytes_per_launch = one["algorithmic_bytes_per_launch"] if one else 0.0
        traffic = one["traffic"] if one else None
        valu = one["valu"] if one else None
        # the INSTANCE-level bound: all wave-level VALU instructions of one instance (PMC, static) over the step time, against the
        # issue peak and against what this instruction mix can reach (scripts/ubench/enc_rates.hip: carry / 64-bit / multiply
        # forms issue at ~1.7x the time of a plain 32-bit add; the mix of the leaf kernel averages ~1.5x)
        valu_instance = None
        if pmc.get("valu_insts_per_instance"):
            vi = pmc["valu_insts_per_instance"]
            ach = vi / (ms_per_step * 1e-3) / 1e9
            al = pmc.get("leaf_one_alone") or {}
            valu_instance = {"unit": "G wave-instructions/s", "insts_per_instance": vi, "achieved": ach, "peak_nominal": issue_peak,
                             "frac_nominal": ach / issue_peak,
                             "source": "profiles/%s (static: rocprofv3 --pmc SQ_INSTS_VALU of this command)" % PMC_FILE}
            if al.get("eff_clock_ghz"):
                # the instance's mix priced like its dominant kernel's (cycles per instruction and clock of poseidon_leaves alone; the
                # transform kernels pair 15 - 18 % of their instructions, the hash kernels 0 - 4 %: profiles/PMC_FILE)
                cal = 1024.0 * al["eff_clock_ghz"] / al.get("cycles_per_inst", 4.0)
                valu_instance.update({"peak_calibrated": cal, "frac_of_calibrated": ach / cal, "eff_clock_ghz": al["eff_clock_ghz"],
                                      "cycles_per_inst": al.get("cycles_per_inst", 4.0)})
        out = {
            "metric": "SIPP proof-gen wall-clock + pairings-aggregated/sec, n=%d (the 3 STARK sub-proofs, %s; outer plonky2 proof not included)"
                      % (args.n, "hardened G1/G2 AIRs" if hardened else "plain G1/G2 AIRs: forgeable for crafted statements, see hardened_instance"),
            "value": value, "unit": "pairings/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            # the three proofs of the last timed step, on every rank, through oracle/stark.c's verifier after the timed region
            "verified": verified_main,
            # ... and through the library's own verifier first (sipp_stark_verify: host C++, one core; ms per proof, G1 / G2 / Fq12)
            "library_verifier": {"ms_per_proof": lib_verify_ms, "entry_point": "sipp_stark_verify (verify_stark_proof of the generators / data.verify)"},
            **({"verify_errors": VERIFY_ERRORS[:3]} if VERIFY_ERRORS else {}),
            # what carried the barrier and the max-over-ranks reduction of the timed region
            "timing_reduction": ("%s:%s" % (dist.get_backend(), red_device)) if dist.is_initialized() else "single process",
            "config": {"workload": "n=%d SIPP instance per GPU: G1ExpStark %d IO (N=2^%d, W+P+Q=%d), G2ExpStark %d IO "
                                   "(N=2^%d, %d), Fq12ExpStark %d IO (N=2^%d, %d); IO records on host -> 3 flat proofs "
                                   "on host, the 3 sub-proofs on 3 concurrent HIP streams" % (args.n, ios[0].shape[0], shapes[0][0], sum(shapes[0][1:]),
                                               ios[1].shape[0], shapes[1][0], sum(shapes[1][1:]),
                                               ios[2].shape[0], shapes[2][0], sum(shapes[2][1:])),
                       "stark_config": "rate_bits=1 cap_height=4 pow_bits=16 arity=16 queries=84 challenges=2 pow_rule=duplex",
                       "air_revision": air_revision(),
                       "air_variant": ("hardened G1 / G2 exponentiation AIRs (API kinds 4 / 5: canonical x3, x-inequality witness, R = +-P proved; "
                                       "+14 % columns) + Fq12ExpStark (kind 2)") if hardened else
                                      "plain G1 / G2 exponentiation AIRs (kinds 0 / 1: incomplete chord rows as recalled from upstream) + Fq12ExpStark",
                       "kinds": kinds,
                       "columns": {"G1ExpStark": {"log_n": shapes[0][0], "W": shapes[0][1], "P": shapes[0][2], "Q": shapes[0][3]},
                                   "G2ExpStark": {"log_n": shapes[1][0], "W": shapes[1][1], "P": shapes[1][2], "Q": shapes[1][3]},
                                   "Fq12ExpStark": {"log_n": shapes[2][0], "W": shapes[2][1], "P": shapes[2][2], "Q": shapes[2][3]}},
                       "parallelism": "%d independent SIPP instance(s), one per GPU, no collective" % world},
            "roofline": {"bound": "hbm", "kernel": "poseidon_leaves", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": ("profiles/%s (static: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                            "command, FETCH_SIZE x2 gfx950 correction; not re-measured in this run)" % PMC_FILE)
                                           if traffic is not None else None,
                         "pmc_counters": pmc_state,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "launches": launches, "avg_launch_ms": avg_ms, "valu": valu, "valu_instance": valu_instance,
                         "two_lane_kernel": pair if pair is not one else None,
                         "note": "one-state-per-lane launches only (%d per step); integer-VALU-bound kernel (Poseidon x^7 + MDS, "
                                 "~%.2f G permutations/s); HBM fraction is small by construction, see DESIGN.md"
                                 % (launches // max(1, args.steps), (one["perms_per_s"] / 1e9) if one else 0.0)},
            # HIP-event brackets on each proof's own stream, summed over the three CONCURRENT streams: upper bounds on a kernel's
            # cost (its waves share the SIMDs with the other proofs' kernels), their sum exceeds ms_per_step
            "kernel_ms_per_step": {k: round(v["ms"] / args.steps, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])},
            "kernel_ms_per_step_note": "per-stream HIP-event time, 3 streams concurrent (sum > ms_per_step)",
            # the same kernels with the three proofs run one after the other (one extra step outside the timed region)
            "kernel_ms_serial": ({k: round(v, 3) for k, v in sorted(kernel_ms_serial.items(), key=lambda kv: -kv[1])}
                                 if kernel_ms_serial else None),
            "roofline_ntt": ntt_roofline(shapes, kernel_ms_serial, pmc),
            # committed LDE cells (2N (W + P + Q) per STARK) per second of wall clock: an AIR-independent rate
            "lde_cells_per_s": sum(2.0 * (1 << s[0]) * (s[1] + s[2] + s[3]) for s in shapes) / (ms_per_step * 1e-3),
            # compulsory HBM traffic of a whole STARK, 8 N (12 W + 12 P + 7 Q) bytes, over the step time
            "stark_bytes_alg_GBs": sum(8.0 * (1 << s[0]) * (12 * s[1] + 12 * s[2] + 7 * s[3]) for s in shapes) / (ms_per_step * 1e-3) / 1e9,
            # companion figures of SURVEY.md section 8(d): leaf permutations and NTT butterflies per second of kernel time
            # (HIP-event time of those kernels, which run concurrently with the other proofs' kernels)
            "rates": {
                "poseidon_leaf_perms_per_s": (leaf_perms * args.steps / ((lk["ms"] + (lkp["ms"] if pair is not one else 0.0)) * 1e-3)
                                              if lk["ms"] > 0 else None),
                # butterflies of all committed transforms over the event time of every NTT / LDE kernel (3 streams concurrent)
                "ntt_butterflies_per_s": (
                    sum((s[1] + s[2]) * ((1 << s[0]) / 2 * s[0] + (1 << s[0]) * (s[0] + 1))
                        + (2 + s[3]) * (1 << s[0]) * (s[0] + 1) for s in shapes) * args.steps
                    / (sum(prof.get(k, {"ms": 0})["ms"] for k in NTT_KERNELS) * 1e-3)
                    if sum(prof.get(k, {"ms": 0})["ms"] for k in NTT_KERNELS) > 0 else None),
            },
            "proof_words": [int(len(p)) for p in proofs],
            "proof_ms_per_step": [round(x / args.steps, 2) for x in proof_ms] if serial else None,
            "pipelined": pipelined,
        }
        if pipelined and valu_instance and valu_instance.get("peak_calibrated"):
            # five instances queued: the same instruction count over the queued time per instance
            pipelined["valu_frac_of_calibrated"] = (valu_instance["insts_per_instance"] / (pipelined["ms_per_instance"] * 1e-3) / 1e9
                                                    / valu_instance["peak_calibrated"] / world)
        # secondary, outside the timed region: the native SIPP chain in front of the circuit (HISTORY.md section 7)
        # on this GPU -- sipp_prove_native = 3n - 2 pairings + the folds, sipp_verify_native = the obligation lists the timed
        # region consumes.  Never allowed to break the main line.
        try:
            st = np.load(os.path.join(ROOT, "tests", "golden", "sipp_n%d_ios.npz" % args.n))["statement"]
            An, Bn = st[: 16 * args.n].reshape(args.n, 16), st[16 * args.n: 48 * args.n].reshape(args.n, 32)
            nctx = ctxs[1]                      # the G2 ctx: its arena holds the widest fold
            nctx.prove_native(An, Bn)
            t = time.perf_counter()
            npf = nctx.prove_native(An, Bn)
            t_np = time.perf_counter() - t
            t = time.perf_counter()
            okn, _, nios = nctx.verify_native(An, Bn, npf)
            t_nv = time.perf_counter() - t
            cpu_pair = None
            if not args.no_cpu_baseline:
                # the CPU restatement of the pairing (oracle/py, big-int Python: a port, not the reference's Rust) on 3 pairs
                from oracle.py import bn254 as obn
                t = time.perf_counter()
                for k in (3, 5, 7):
                    obn.pairing(obn.g1_mul(obn.G1, k), obn.g2_mul(obn.G2, k + 1))
                cpu_pair = {"pairings_per_s": 3 / (time.perf_counter() - t), "cores": 1, "kind": "port",
                            "sample": "oracle/py/bn254.py pairing (pure Python big integers) on 3 pairs"}
            out["native_chain"] = {"cpu_baseline": cpu_pair, "prove_native_ms": 1e3 * t_np, "verify_native_ms": 1e3 * t_nv, "pairings": 3 * args.n - 2,
                                   "pairings_per_s": (3 * args.n - 2) / t_np, "accepted": bool(okn),
                                   "verified": bool(okn) and bool(all((a == b).all() for a, b in zip(nios, ios))),
                                   "obligations_equal_bench_input": bool(all((a == b).all() for a, b in zip(nios, ios)))}
        except Exception as e:                  # noqa: BLE001
            out["native_chain"] = {"error": repr(e)}
        # secondary, outside the timed region: the messages -> G2 step in front of SIPP's BLS example (reference
        # src/bin/bls_aggregation.rs:65, :100-104; HISTORY.md section 7b) for the n - 1 messages of this instance: the native map with
        # its cofactor clearing, and the MapToG2 proof (eight trace rows per message).  The 2 (n - 1) cofactor obligations are ordinary
        # G2ExpStark records (the main line's path), not timed again here.
        try:
            if os.environ.get("SIPP_BENCH_MAP_G2", "1") in ("0", ""):      # scripts/profile_round.sh: counters of the instance alone
                raise RuntimeError("skipped (SIPP_BENCH_MAP_G2=0)")
            mctx = ctxs[1]
            rng = np.random.default_rng(0x6d6170)
            msgs = np.zeros((args.n - 1, 16), dtype=np.uint32)
            msgs[:, :7] = rng.integers(0, 2**32, size=(args.n - 1, 7), dtype=np.uint64)       # u.c0 < 2^224 < p
            msgs[:, 8:15] = rng.integers(0, 2**32, size=(args.n - 1, 7), dtype=np.uint64)
            mctx.map_to_g2(msgs)
            t = time.perf_counter()
            mrec, mg2, mpts = mctx.map_to_g2(msgs)
            t_map = time.perf_counter() - t
            mctx.prove(3, mrec)
            t = time.perf_counter()
            mproof = mctx.prove(3, mrec)
            t_mp = time.perf_counter() - t
            out["map_to_g2"] = {"messages": args.n - 1, "map_and_cofactor_ms": 1e3 * t_map, "proof_ms": 1e3 * t_mp,
                                "verified": verify_proofs([mproof]),
                                "proof_words": int(len(mproof)), "shape": list(mctx.shape(3, args.n - 1)),
                                "cofactor_obligations": int(mg2.shape[0]),
                                "entry_points": "sipp_map_to_g2, sipp_map_to_g2_prove"}
        except Exception as e:                  # noqa: BLE001
            out["map_to_g2"] = {"error": repr(e)}
        # secondary, outside the timed region: the in-circuit FINAL PAIRING of the BLS example (src/bin/bls_aggregation.rs:76-77) as a STARK
        # obligation: the record (final_A, final_B, final_Z) of THIS instance's statement, one pairing in 2^13 trace rows (kind 6)
        try:
            if os.environ.get("SIPP_BENCH_PAIRING", "1") in ("0", ""):
                raise RuntimeError("skipped (SIPP_BENCH_PAIRING=0)")
            st = np.load(os.path.join(ROOT, "tests", "golden", "sipp_n%d_ios.npz" % args.n))["statement"]
            prec = np.ascontiguousarray(st[-144:].reshape(1, 144))
            pctx = sipp_amd.Ctx(device=local_rank, workspace_bytes=sipp_amd.lib().sipp_workspace_bytes(6, 1))
            try:
                pctx.prove(6, prec)
                pctx.profile(True)
                pctx.profile_reset()
                t = time.perf_counter()
                for _ in range(5):
                    pproof = pctx.prove(6, prec)
                t_pp = (time.perf_counter() - t) / 5
                prep = pctx.profile_report()
                pctx.profile(False)
                out["final_pairing"] = {"records": 1, "proof_ms": 1e3 * t_pp, "verified": verify_proofs([pproof]), "proof_words": int(len(pproof)),
                                        "shape": list(pctx.shape(6, 1)),
                                        "kernel_ms_per_proof": {k: round(v["ms"] / 5, 3) for k, v in sorted(prep.items(), key=lambda kv: -kv[1]["ms"])[:8]},
                                        "entry_point": "sipp_pairing_prove (pairing_circuit(final_A, final_B) == final_Z as a STARK obligation)"}
            finally:
                pctx.close()
        except Exception as e:                  # noqa: BLE001
            out["final_pairing"] = {"error": repr(e)}
        # secondary, outside the timed region: the same instance with the OTHER AIR variant of G1 / G2 -- by default the plain kinds 0 / 1
        # (this repository's reading of the recalled upstream AIR, starky-bn254's incomplete chord rows: forgeable for crafted
        # statements; 14 % fewer columns), next to the headline that proves the hardened kinds 4 / 5.  Three
        # ctxs of their own, through sipp_instance_prove, sipp_amd.Instance's default stream levels for that variant.
        other_key = "recalled_upstream_air" if hardened else "hardened_instance"
        try:
            if os.environ.get("SIPP_BENCH_OTHER_AIR", "1") in ("0", ""):
                raise RuntimeError("skipped (SIPP_BENCH_OTHER_AIR=0)")
            oinst = sipp_amd.Instance([a.shape[0] for a in ios], devices=(local_rank,) * 3, hardened=not hardened)
            try:
                oinst.prove(ios)
                t = time.perf_counter()
                for _ in range(10):
                    hp = oinst.prove(ios)
                t_h = (time.perf_counter() - t) / 10
                out[other_key] = {"kinds": [int(x[1]) for x in hp], "ms_per_instance": 1e3 * t_h, "value": args.n / t_h, "unit": "pairings/s",
                                  "verified": verify_proofs(hp),
                                  "columns": [list(oinst.ctxs[i].shape(i, ios[i].shape[0])) for i in range(3)],
                                  "proof_words": [int(len(x)) for x in hp],
                                  "entry_point": "sipp_instance_prove" + ("" if hardened else " on ctxs with sipp_ctx_set_hardened"),
                                  "note": ("plain G1 / G2 AIRs: where the accumulator meets the running power on an add row the chord rule leaves "
                                           "the slope free (tests/test_oracle_hardened.py builds the forgery); not the default since round 4")
                                          if hardened else "hardened G1 / G2 AIRs (kinds 4 / 5)"}
            finally:
                oinst.close()
        except Exception as e:                  # noqa: BLE001
            out[other_key] = {"error": repr(e)}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(ios, shapes, kinds)
            out["cpu_baseline"]["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        else:
            out["cpu_baseline"] = None
    # secondary figure, outside the timed region and after everything else (this process's other arenas are released first: at
    # n = 4096 one rank needs 246 GB): ONE instance whose obligation lists are cut into `world` contiguous ranges, every rank
    # proving its range as STARKs of its own (sipp_io_shard; DESIGN.md section 5, level L-D; BASELINE configs[3] / configs[4]:
    # "per-round starky sub-proofs sharded across 8 MI355X").  Strong scaling, no data-path collective: the time of the
    # instance is the slowest rank's.  Sizes: n = 1024 and n = 4096 on any number of ranks (one rank = the whole instance on one
    # GPU, the figure the sharded runs are divided into).
    inst.close()
    io_sharded = None
    shard_sizes = os.environ.get("SIPP_BENCH_IO_SHARD_N", "1024,4096")
    if shard_sizes not in ("", "0") and not serial:
        io_sharded = {}
        for n_s in [int(x) for x in shard_sizes.split(",")]:
            mine = sipp_amd.shard_ios(load_ios(n_s), world, rank)
            si = sipp_amd.Instance([a.shape[0] for a in mine], devices=(local_rank,) * 3, priorities=prios, hardened=hardened)
            k_s = 2
            si_single = si.single_ctx
            arena_bytes = sum(int(c.workspace_bytes) for c in si.distinct_ctxs())
            own = []
            dts, sproofs = dist_util.timed_steps(lambda: si.prove(mine), k_s, 1, sync=lambda: (torch.cuda.synchronize(), si.sync()),
                                                 device=red_device, local_out=own)
            sproofs = [p.copy() for p in sproofs]
            si.close()
            s_ok = all_ranks_true(verify_proofs(sproofs), red_device)     # every rank's shard proofs of the last step, after the timing
            lo, hi = dist_util.min_max_over_ranks(own[0], device=red_device)
            io_sharded["n=%d" % n_s] = {"ms_per_instance": 1e3 * dts / k_s, "value": n_s * k_s / dts, "unit": "pairings/s",
                                        # one rank proves the whole instance on one GPU: the figure the sharded runs are divided into
                                        "scaling": "strong" if world > 1 else "baseline (whole instance on one GPU)",
                                        "ranks": world, "steps": k_s, "verified": s_ok,
                                        "rank_ms_per_instance_min_max": [1e3 * lo / k_s, 1e3 * hi / k_s],
                                        "host_threads_per_rank": 1 if si_single else 4,
                                        # True: one ctx / one arena, the three proofs back to back (the three arenas together would
                                        # exceed 90 % of the card's memory, or what is free of it: n = 4096 on one rank)
                                        "single_ctx": si_single, "kinds": kinds,
                                        # HBM this rank reserved for its shard (the sum of its ctxs' arenas)
                                        "arena_bytes_per_rank": arena_bytes,
                                        # what a `world`-GPU run of this instance is PROJECTED to take, from one GPU proving every rank's shard alone
                                        # (scripts/io_shard_projection.py, committed): no multi-GPU run has been possible on this pool -- every
                                        # speed-up figure of DESIGN.md section 5 is such a projection
                                        "projected_world8": projected_shards(n_s, hardened),
                                        "records_of_rank0": [int(a.shape[0]) for a in mine],
                                        # what this sharding costs the verifier side (sipp_amd/proof_cost.py, counted from the proofs'
                                        # shapes): 3 * world proofs instead of 3 for the recursive verifier behind verifier_circuit.rs:133-147
                                        "verifier_price": price_vs_world1([a.shape[0] for a in load_ios(n_s)], world, hardened)}
    if rank == 0:
        out["io_sharded"] = io_sharded
        # secondary, one rank only: the outer plonky2 prover at the reference's circuit configuration (SIPP_BENCH_OUTER_PLONK=0 skips,
        # SIPP_BENCH_OUTER_PLONK=<log2 rows> sizes it; default 2^18)
        op_bits = os.environ.get("SIPP_BENCH_OUTER_PLONK", "18")
        if world == 1 and op_bits not in ("", "0") and not serial:
            try:
                out["outer_plonk"] = outer_plonk_leg(local_rank, int(op_bits))
            except Exception as e:                  # noqa: BLE001 -- never allowed to break the main line
                out["outer_plonk"] = {"error": repr(e)}
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
