/*
 * sipp_hip.h -- C ABI of the MI355X-native SIPP STARK sub-prover (libsipp_hip.so).
 *
 * This is the drop-in boundary for the hot path of qope/SIPP: the three starky
 * sub-proofs that reference src/verifier_circuit.rs:133-135 delegates to
 * starky-bn254 (g1_exp_circuit / g2_exp_circuit / fq12_exp_circuit).  At proving
 * time each of those runs, inside a plonky2 witness generator,
 *     stark.generate_trace(&ios); stark.generate_public_inputs(&ios);
 *     starky::prover::prove::<F, C, S, D>(stark, &config, trace, public_inputs, &mut timing)
 * (starky-bn254 @ 2d46f9e and starky @ InternetMaximalism/plonky2 541e127,
 * reference Cargo.toml:24,26 -- not vendored under /root/reference).  The three
 * sipp_*_exp_prove entry points below replace exactly that body: native IO
 * records in, flat StarkProofWithPublicInputs buffer out.  Trace fill happens
 * on the device; the caller never materialises the N x W table.
 *
 * Conventions
 *  - plain C, no callbacks, no exceptions; every function returns a sipp_status.
 *  - field elements are canonical little-endian uint64_t (< 2^64 - 2^32 + 1);
 *    extension elements are two consecutive uint64_t (c0, c1).
 *  - BN254 integers are 8 x uint32_t little-endian limbs, exactly the limb
 *    stream of reference src/transcript_native.rs:68-77 and src/statements.rs:90-131.
 *  - `d_` parameters are DEVICE pointers (HBM of the ctx's GPU); everything else
 *    is host memory owned by the caller.
 *  - a sipp_ctx is bound to one GPU and is not thread-safe; distinct contexts on
 *    distinct GPUs may be used concurrently (one process per GPU in bench.py).
 */
#ifndef SIPP_HIP_H
#define SIPP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sipp_ctx sipp_ctx;

typedef enum {
    SIPP_OK = 0,
    SIPP_E_BADARG = -1,      /* null pointer, non power of two, size out of range */
    SIPP_E_HIP = -2,         /* a HIP runtime call failed; see sipp_last_error */
    SIPP_E_NOMEM = -3,       /* workspace arena exhausted */
    SIPP_E_BUFSZ = -4,       /* caller's output buffer too small */
    SIPP_E_SUBGROUP = -5,    /* zeta landed in the trace subgroup (starky prover.rs ensure!) */
    SIPP_E_QUOTIENT = -6,    /* constraints not satisfied: quotient has too high degree */
    SIPP_E_UNSUPPORTED = -7, /* configuration not supported by this build */
    SIPP_E_WITNESS = -8,     /* IO record not provable (point at infinity / not on curve) */
    SIPP_E_VERIFY = -9       /* sipp_stark_verify: the proof is not accepted (the stage is reported through `reason`) */
} sipp_status;

/* starky StarkConfig::standard_fast_config(); supported ranges in brackets */
typedef struct {
    uint32_t rate_bits;       /* 1   [1 .. 3: blowup 2, 4, 8; the quotient stays on the 2N coset]   */
    uint32_t cap_height;      /* 4   [0 .. 8]                                                        */
    uint32_t pow_bits;        /* 16  [0 .. 32]                                                       */
    uint32_t arity_bits;      /* 4   [1 .. 4: FriReductionStrategy::ConstantArityBits, arity 2 .. 16] */
    uint32_t final_poly_bits; /* 5   [0 .. 12]; together with arity_bits it must leave every committed FRI layer at least 16
                                 values for the trace length proved (else SIPP_E_UNSUPPORTED from the prove call)   */
    uint32_t num_queries;     /* 84  [1 .. 1024]                                                     */
    uint32_t num_challenges;  /* 2   [2]                                                             */
    uint32_t pow_rule;        /* SIPP_POW_DUPLEX (0): observe the witness, response = next challenge (plonky2 fri/prover.rs
                                 of 2023); SIPP_POW_HASH (1): response = hash_no_pad(challenger.get_hash() || witness)[0]
                                 (the earlier rule).  Kept as data until upstream's pinned revision can be read. */
    uint32_t fs_rule;         /* where Fiat-Shamir starts.  SIPP_FS_STATEMENT (0, default): this repository's format -- the challenger
                                 observes the statement (kind, shape, config, Merkle root of the public inputs) before the trace cap;
                                 SIPP_FS_UPSTREAM (1): starky's order as recalled: the challenger starts at the
                                 trace cap, the public inputs are NOT in the transcript (upstream's exposure: a verifier must bind them
                                 by other means, as the recursive circuit does by connecting them to its own targets) */
    uint32_t lookup_rule;     /* challenges of a lookup's two permutation factors.  SIPP_LOOKUP_INDEPENDENT (0, default):
                                 Z' (pin + gamma)(ptab + beta) = Z (col + gamma)(tab + beta) -- column and table cannot trade entries;
                                 SIPP_LOOKUP_SHARED (1): both factors use gamma, (beta, gamma) are drawn as before and beta is unused
                                 -- starky's permutation pairs of ONE column each as recalled (reduce_with_powers over a single
                                 column leaves beta^0), which proves only the union multiset.  Header word 15 of a flat proof
                                 records fs_rule | lookup_rule << 1. */
} sipp_stark_config;
#define SIPP_POW_DUPLEX 0
#define SIPP_POW_HASH 1
#define SIPP_FS_STATEMENT 0
#define SIPP_FS_UPSTREAM 1
#define SIPP_LOOKUP_INDEPENDENT 0
#define SIPP_LOOKUP_SHARED 1

void sipp_default_config(sipp_stark_config *cfg);

/* Which STARK (reference src/verifier_circuit.rs:133 / :134 / :135) */
typedef enum { SIPP_G1_EXP = 0, SIPP_G2_EXP = 1, SIPP_FQ12_EXP = 2,
               SIPP_MAP_G2 = 3 /* src/bin/bls_aggregation.rs:65, see sipp_map_to_g2_prove */,
               /* G1 / G2 exponentiation with the HARDENED AIR (same records, same rows, about 14 % more columns): x3 canonical; where
                * an addition is used, an inequality witness for the two x-coordinates -- the chord rule of the plain AIR is satisfied by
                * ANY slope where the accumulator meets the running power; that case (R = P) is proved through the
                * next row's double and R = -P through an identity state bit instead of being refused: every record whose OUTPUT is a
                * finite point has a proof, except one that meets R = P on the LAST add row of its 512 rows (bit 255 of a 256-bit exponent:
                * no row is left to hand the double over) -- SIPP_E_WITNESS, like an output at the identity.  Through sipp_prove / sipp_prove_async and the generic size / shape / trace functions. */
               SIPP_G1_EXP_HARDENED = 4, SIPP_G2_EXP_HARDENED = 5,
               SIPP_PAIRING = 6 /* src/bin/bls_aggregation.rs:76-77, see sipp_pairing_prove */ } sipp_kind;

/* u32 words per IO record, (x, offset, exp_val, output) order:
 * G1 7*8 = 56, G2 13*8 = 104, Fq12 37*8 = 296. */
#define SIPP_G1_IO_WORDS 56
#define SIPP_G2_IO_WORDS 104
#define SIPP_FQ12_IO_WORDS 296
/* MapToG2 records (u, x, y): the message in Fp2 and its point on the twist, 6*8 = 48 */
#define SIPP_MAP_G2_IO_WORDS 48
/* final-pairing records (P, Q, Z): a G1 point (16), a G2 point (32: x.c0, x.c1, y.c0, y.c1), e(P, Q) as 12 MyFq12 coefficients (96) */
#define SIPP_PAIRING_IO_WORDS 144

/* ---- context ---------------------------------------------------------------- */
/* `workspace_bytes` of HBM are reserved once; afterwards the only allocations are the constant tables of a transform SIZE at its
 * first use on the ctx (twiddles / node constants: at most 8 B per LDE row, built on the host once and kept until destroy).
 * Pass 0 to size the arena for n = 128 (about 24 GiB). */
int sipp_ctx_create(sipp_ctx **out, int device, const sipp_stark_config *cfg, size_t workspace_bytes);
/* The version of THIS header (bumped whenever a struct of the ABI changes size or meaning: 2 = sipp_stark_config with ten fields,
 * round 5; 3 = round 6, SIPP_PAIRING).  A binding that cannot be rebuilt with the library (the Rust shim) creates its contexts through
 * sipp_ctx_create_checked, which refuses (SIPP_E_BADARG, *out = NULL, nothing read from cfg) a caller whose header version or
 * sizeof(sipp_stark_config) differs from the library's -- a struct two fields short would otherwise be read past its end and stray
 * bits taken for fs_rule / lookup_rule. */
#define SIPP_ABI_VERSION 3u
uint32_t sipp_abi_version(void);
int sipp_ctx_create_checked(sipp_ctx **out, int device, const sipp_stark_config *cfg, size_t cfg_size, uint32_t abi_version,
                            size_t workspace_bytes);
/* Which of two equivalent kernels serves a step (results are identical; the non-default routes are the fallbacks the tests keep
 * alive -- until round 5 they hid behind environment variables).  routes = OR of:
 *   SIPP_ROUTE_OPENINGS_UNGROUPED  openings at zeta / g zeta by one block per column instead of the grouped kernel (traces >= 1024 rows)
 *   SIPP_ROUTE_LDE_COLUMN_WIDE     columns of 2^13 / 2^14 rows through the whole-column-in-LDS transform instead of the tree sweeps
 *   SIPP_ROUTE_WITNESS_NO_GRAPH    sipp_plonk_generate_witness_levels launches its two kernels per level one by one instead of replaying
 *                                  the captured hipGraph
 * Only while no proof is in flight on the ctx. */
#define SIPP_ROUTE_OPENINGS_UNGROUPED 1u
#define SIPP_ROUTE_LDE_COLUMN_WIDE 2u
#define SIPP_ROUTE_WITNESS_NO_GRAPH 4u
int sipp_ctx_set_kernel_routes(sipp_ctx *ctx, uint32_t routes);
void sipp_ctx_destroy(sipp_ctx *ctx);
/* The ctx's HIP stream: level > 0 = a stream of the highest priority the device offers; level <= 0 = a stream with a hardware
 * queue of its own at normal priority (created with an all-ones CU mask: the runtime multiplexes ordinary streams onto a pool of
 * four queues, such a stream is kept out of the pool -- measured 1.5 ms per n = 128 instance, HISTORY.md section 6c; level < 0 used
 * to mean the lowest priority and still does under SIPP_DEDICATED_QUEUES=0).  Such a stream is BLOCKING with respect to the legacy
 * default (NULL) stream -- hipExtStreamCreateWithCUMask takes no flags -- whereas the high-priority stream is hipStreamNonBlocking:
 * an embedder that launches on the NULL stream serialises with the level <= 0 ctxs (not with a level > 0 one); launch on streams
 * of your own (or per-thread default streams), or run with SIPP_DEDICATED_QUEUES=0 (every ctx stream non-blocking, the runtime's
 * queue pool, about 1 ms per n = 128 instance).  With the three proofs of an instance on three ctxs:
 * G1 and G2 <= 0, Fq12 high.  Only while no proof is in flight on the ctx.  sipp_ctx_create makes a level-0 stream. */
int sipp_ctx_set_stream_priority(sipp_ctx *ctx, int level);
/* on != 0: on this ctx the kinds SIPP_G1_EXP / SIPP_G2_EXP mean the HARDENED AIRs (SIPP_G1_EXP_HARDENED / SIPP_G2_EXP_HARDENED below) in
 * every call that takes the ctx -- sipp_g1_exp_prove, sipp_prove_async, sipp_instance_prove, sipp_instances_prove, sipp_proof_size,
 * sipp_stark_shape, sipp_trace_build; the proof's header carries kind 4 / 5.  Size the arena with sipp_workspace_bytes(kind + 4, ..)
 * (that function takes no ctx).  Only while no proof is in flight on the ctx. */
int sipp_ctx_set_hardened(sipp_ctx *ctx, int on);
const char *sipp_last_error(const sipp_ctx *ctx);
int sipp_sync(sipp_ctx *ctx);
/* the hipStream_t every launch of this ctx goes to (as void*) */
void *sipp_stream(sipp_ctx *ctx);

/* ---- the three provers (replace the body of starky-bn254's *ProofGenerator::run_once) */
/* ios: num_io records of SIPP_*_IO_WORDS u32 each, outputs included (they are
 * checked against the device-computed result).  num_io is padded internally to
 * the next power of two by repeating the last record, as upstream does.
 * On success *proof_len u64 words of proof_out are written (layout: INTEGRATION.md). */
int sipp_g1_exp_prove(sipp_ctx *ctx, const uint32_t *ios, size_t num_io, uint64_t *proof_out, size_t proof_cap,
                      size_t *proof_len);
int sipp_g2_exp_prove(sipp_ctx *ctx, const uint32_t *ios, size_t num_io, uint64_t *proof_out, size_t proof_cap,
                      size_t *proof_len);
int sipp_fq12_exp_prove(sipp_ctx *ctx, const uint32_t *ios, size_t num_io, uint64_t *proof_out, size_t proof_cap,
                        size_t *proof_len);
/* The same for any sipp_kind (the only synchronous entry point of the hardened kinds). */
int sipp_prove(sipp_ctx *ctx, int kind, const uint32_t *ios, size_t num_io, uint64_t *proof_out, size_t proof_cap,
               size_t *proof_len);
/* Asynchronous form ("who calls it"): plonky2 runs the three witness generators behind
 * reference src/verifier_circuit.rs:133-135 serially on one thread, so a patched caller starts each proof as soon as
 * its IO values exist and collects the three afterwards.  sipp_prove_async hands the job to the ctx's own worker
 * thread and returns at once; `ios` and `proof_out` must stay valid until sipp_wait returns.  One proof in flight
 * per ctx (a second sipp_prove_async before sipp_wait fails with SIPP_E_BADARG).  sipp_wait blocks until the
 * proof is complete and returns ITS status; *proof_len as for the synchronous calls. */
int sipp_prove_async(sipp_ctx *ctx, int kind, const uint32_t *ios, size_t num_io, uint64_t *proof_out,
                     size_t proof_cap);
int sipp_wait(sipp_ctx *ctx, size_t *proof_len);
/* One SIPP instance = the three sub-proofs, concurrently on three DISTINCT ctxs (one HIP stream each; they may sit
 * on one GPU or on up to three GPUs, SURVEY.md section 8e level L-B).  Arrays are indexed by sipp_kind.  Returns the
 * first failing status; every proof that was started is waited for in any case.  num_io[k] == 0 skips kind k
 * (proof_len[k] = 0).
 * ctxs[0] == ctxs[1] == ctxs[2]: ONE ctx, the three proofs back to back on its stream and arena (workspace = the largest of the
 * three sipp_workspace_bytes) -- for configurations whose three arenas do not fit the card together: n = 4096 needs 159 GB this
 * way instead of 246 GB (179 / 276 GB with the hardened AIRs, which only fit this way) at 5.5 % more time per instance (1234 against
 * 1169 ms; 13 - 15 % at n = 1024, where the proofs' thin phases overlap more).  Two equal handles and a third are SIPP_E_BADARG. */
int sipp_instance_prove(sipp_ctx *const ctxs[3], const uint32_t *const ios[3], const size_t num_io[3],
                        uint64_t *const proof_out[3], const size_t proof_cap[3], size_t proof_len[3]);
/* A queue of `count` independent instances on one GPU (or several): `in_flight` slots of three distinct ctxs each
 * (ctxs[3 * slot + kind]) take the instances from a shared counter, so the latency-bound head and tail of one instance
 * overlap the hashing of the others.  Arrays of instance i: ios[3 i + kind], num_io[3 i + kind], proof_out / proof_cap /
 * proof_len likewise; status[i] (may be NULL) receives sipp_instance_prove's result for instance i.  Returns the first
 * failing status; every instance is attempted.  Every ctx's workspace must hold the largest proof of its kind. */
int sipp_instances_prove(sipp_ctx *const *ctxs, size_t in_flight, size_t count, const uint32_t *const *ios,
                         const size_t *num_io, uint64_t *const *proof_out, const size_t *proof_cap, size_t *proof_len,
                         int *status);
/* IO-sharded sub-proofs (level L-D): the obligation list of one kind is cut into
 * `world` contiguous, balanced ranges and every rank proves ITS range as a STARK of its own -- on the reference side one
 * g1_exp_circuit / g2_exp_circuit / fq12_exp_circuit call per range instead of one per list (src/verifier_circuit.rs:133-135
 * take any slice of obligations).  No data is exchanged between ranks.  Range of `rank`: records [*first, *first + *count);
 * *count may be 0 when num_io < world (sipp_instance_prove skips a kind whose num_io is 0 and reports proof_len 0). */
int sipp_io_shard(size_t num_io, uint32_t world, uint32_t rank, size_t *first, size_t *count);
/* The outputs alone: what the generators behind g1_exp_circuit / g2_exp_circuit / fq12_exp_circuit assign to the
 * returned output targets (reference src/verifier_circuit.rs:133-135 `*_exp_outputs`): out = offset + [exp_val] x
 * (G1, G2) or offset * x^exp_val (Fq12), computed on the device by the trace kernels' accumulator chains.  `ios` are
 * full-size records as for the provers; their (x, offset, exp_val) words are read and their output words are
 * OVERWRITTEN.  Unprovable inputs (a point at infinity on the way) return SIPP_E_WITNESS as in the provers. */
int sipp_exp_outputs(sipp_ctx *ctx, int kind, uint32_t *ios, size_t num_io);
/* upper bound (u64 words) of the flat proof for `num_io` records of `kind` */
size_t sipp_proof_size(const sipp_ctx *ctx, int kind, size_t num_io);
/* HBM workspace (bytes) one proof of `kind` with `num_io` records needs; pass it (or more) to sipp_ctx_create */
size_t sipp_workspace_bytes(int kind, size_t num_io);
/* the same for a non-default configuration (the LDEs grow with the blowup 2^rate_bits); cfg == NULL = default */
size_t sipp_workspace_bytes_cfg(int kind, size_t num_io, const sipp_stark_config *cfg);
/* memory of GPU `device` in bytes (hipMemGetInfo): what is free now and what the card has -- to decide between three concurrent
 * arenas and one (sipp_instance_prove).  Either pointer may be NULL. */
int sipp_device_memory(int device, size_t *free_bytes, size_t *total_bytes);
/* trace shape the prover will use: rows (log2), main columns, permutation-Z columns, quotient chunks */
int sipp_stark_shape(const sipp_ctx *ctx, int kind, size_t num_io, uint32_t *log_rows, uint32_t *main_cols,
                     uint32_t *perm_cols, uint32_t *quotient_cols);

/* ---- next row: messages -> G2 in front of SIPP's BLS example ----------------- */
/* The STARK behind `batch_map_to_g2_circuit(builder, &messages)` (reference src/bin/bls_aggregation.rs:65): for every
 * message u in Fp2 the point (x, y) = map_to_g2_without_cofactor_mul(u) on the twist y^2 = x^3 + 3/(9+u) (:102; the
 * Shallue - van de Woestijne map of RFC 9380 F.1 with Z = 1, sgn0(y) = sgn0(u)).  ios: num_io records (u, x, y) of
 * SIPP_MAP_G2_IO_WORDS u32, the point included (compared with the device-computed one; SIPP_E_WITNESS if it differs or if a
 * word is >= p; the four u with u^2 g(1) = +-1 follow the RFC's inv0(0) = 0).  Eight trace rows per message, at least 1024 rows.  The cofactor
 * multiplication that follows the map in the reference (`mul_by_cofactor`, :103) is a pair of ordinary G2ExpStark
 * obligations: sipp_map_to_g2 below writes them.  Proof layout and the generic entry points (sipp_prove_async,
 * sipp_proof_size, sipp_workspace_bytes, sipp_stark_shape, sipp_trace_build, sipp_exp_outputs) as for the other kinds
 * with kind = SIPP_MAP_G2. */
int sipp_map_to_g2_prove(sipp_ctx *ctx, const uint32_t *ios, size_t num_io, uint64_t *proof_out, size_t proof_cap,
                         size_t *proof_len);
/* The STARK behind the in-circuit FINAL PAIRING of the reference's BLS example (src/bin/bls_aggregation.rs:76-77:
 * `let z = pairing_circuit(builder, final_A, final_B); Fq12Target::connect(builder, &z, &final_Z)`), the way this path does
 * everything else: as a STARK obligation instead of outer-circuit gates.  ios: num_io records (P, Q, Z) of SIPP_PAIRING_IO_WORDS u32;
 * Z = e(P, Q) with the value `plonky2_bn254_pairing::pairing::pairing` returns (src/prover_native.rs:8,20; as recalled it restates
 * arkworks' Bn254::pairing: the reduced optimal ate pairing raised to 2u(6u^2 + 3u + 1), ark-ec's final exponentiation) is compared with the device-computed one (SIPP_E_WITNESS if it
 * differs, if P / Q are off their curves, if a word is >= p, or if a step of the affine Miller loop degenerates -- Q outside the
 * r-torsion).  2^13 trace rows per pairing, one modular identity per row (64 tangent + 38 chord steps, the easy part, ark-ec's hard-part
 * chain: 7185 active rows), at least two blocks.  A VERIFIER of such a proof must check [r] Q = O besides the curve equations (oracle/stark.c does): the chord
 * rows are sound for points of order r only.  Generic entry points (sipp_prove, sipp_prove_async, sipp_proof_size,
 * sipp_workspace_bytes, sipp_stark_shape, sipp_trace_build, sipp_exp_outputs: Z computed and written) with kind = SIPP_PAIRING. */
int sipp_pairing_prove(sipp_ctx *ctx, const uint32_t *ios, size_t num_io, uint64_t *proof_out, size_t proof_cap,
                       size_t *proof_len);
/* ---- verification ----------------------------------------------------------------------------------------------------------
 * The check of a flat proof made by any of the provers above (every kind): what starky's native `verify_stark_proof` does inside the
 * reference's proof generators right after `prove` (starky-bn254 @ 2d46f9e, un-vendored) and what `data.verify`
 * (src/verifier_circuit.rs:254) rests on.  HOST code (about 40 k Poseidon permutations and one evaluation of the AIR at zeta: a few
 * milliseconds to a few tens on up to eight threads; no GPU, no ctx): header and configuration, canonical words, the PUBLIC conditions of the
 * statement (record elements below p; points on their curves; MapToG2: the sign rule; pairing: Q of order r), the Fiat-Shamir replay,
 * the constraints at zeta against the quotient openings, FRI (proof of work, Merkle paths of the three oracles and of every layer, the
 * fold of every query, the final polynomial).  cfg = NULL: the default configuration.
 * Returns SIPP_OK, SIPP_E_BADARG / SIPP_E_UNSUPPORTED, or SIPP_E_VERIFY with the stage in *reason (may be NULL):
 *   100 header  101 kind / size  102 shape or configuration differs  103 FRI rounds  104 / 105 public-input words  106 truncated
 *   107 zeta in the trace domain  108 a record element >= p  109 a record's point off its curve / sign rule / order of Q
 *   110 / 111 the constraints of challenge 0 / 1 do not meet the quotient  120 FRI section  121 proof of work
 *   122 query truncated  123 / 124 / 125 Merkle path of the trace / Z / quotient oracle  130 layer truncated  131 fold mismatch
 *   132 Merkle path of a layer  133 final polynomial  140 trailing words  141 a word that is not a canonical field element
 * (the oracle's verifier, oracle/stark.c, reports the same stages: tests/test_product_verifier.py compares them proof by proof). */
int sipp_stark_verify(const uint64_t *proof, size_t len, const sipp_stark_config *cfg, int *reason);

/* What the reference computes natively per message (src/bin/bls_aggregation.rs:100-104), on the device:
 *   map_ios [n][48]      (u, x, y) records for sipp_map_to_g2_prove;
 *   g2_ios  [2n][104]    (may be NULL) the G2ExpStark obligations of the cofactor clearing, h = 2p - r:
 *                        record i      : x = (x, y)_i, offset = G2 generator, exp_val = h, output = G + [h](x, y)_i
 *                        record n + i  : x = -G,       offset = that output,  exp_val = 1, output = [h](x, y)_i
 *   cleared [n][32]      (may be NULL) the points [h](x, y)_i of G2 (x.c0, x.c1, y.c0, y.c1), i.e. `ms` of :100-104. */
int sipp_map_to_g2(sipp_ctx *ctx, const uint32_t *msgs, size_t n, uint32_t *map_ios, uint32_t *g2_ios, uint32_t *cleared);

/* ---- next row: the native prover's pairing products ------------------------- */
/* inner_product of reference src/prover_native.rs:15-23: prod_i pairing(A_i, B_i) for n pairs, on the device.
 * g1: n x 16 u32 (x, y), g2: n x 32 u32 (x.c0, x.c1, y.c0, y.c1) -- the limb stream of src/transcript_native.rs:42-54;
 * all-zero coordinates stand for the point at infinity.  out: the 12 MyFq12 coefficients, 96 u32 (src/transcript_native.rs:32-40).
 * sipp_inner_products computes `count` independent products (pairs [k n, (k + 1) n) -> out[k]) in one pass, e.g. the Z_L and
 * Z_R of one SIPP round (src/prover_native.rs:51-52).  Uses the ctx's workspace (about 500 bytes per pair). */
int sipp_inner_product(sipp_ctx *ctx, const uint32_t *g1, const uint32_t *g2, size_t n, uint32_t *out);
int sipp_inner_products(sipp_ctx *ctx, const uint32_t *g1, const uint32_t *g2, size_t n, size_t count, uint32_t *out);

/* The native chain around those products (host code over the device routines; the transcript stays on the host):
 * sipp_prove_native = reference src/prover_native.rs:26-80: A (n x 16 u32), B (n x 32 u32), n a power of two ->
 *   proof = 2 log2 n + 1 Fq12 messages of 96 u32 each, in the reference's (reversed) order; sipp_native_proof_words(n) u32.
 * sipp_verify_native = reference src/verifier_native.rs:14-85: replays the transcript, folds A, B, Z and returns
 *   statement (16n + 32n + 96 + 16 + 32 + 96 u32: A | B | Z | final_A | final_B | final_Z, src/statements.rs:24-39),
 *   the three obligation lists as complete IO records for sipp_*_exp_prove (src/verifier_circuit.rs:68-131:
 *   (n - 1) x 56, (n - 1) x 104 and 2 log2 n x 296 u32, round-major), and *accepted = (pairing(final_A, final_B) == final_Z).
 *   Any output pointer may be NULL.  The ctx's workspace must hold the widest fold (sipp_workspace_bytes(SIPP_G2_EXP, n / 2)). */
size_t sipp_native_proof_words(size_t n);
int sipp_prove_native(sipp_ctx *ctx, const uint32_t *A, const uint32_t *B, size_t n, uint32_t *proof);
int sipp_verify_native(sipp_ctx *ctx, const uint32_t *A, const uint32_t *B, size_t n, const uint32_t *proof,
                       uint32_t *statement, uint32_t *g1_ios, uint32_t *g2_ios, uint32_t *fq12_ios, int *accepted);

/* ---- next row (SURVEY.md section 8f, rank 2 -- the generic half of the outer plonky2 prover) ---------------------------
 * PolynomialBatch commitments and FRI opening proofs for ARBITRARY FriParams: what reference src/verifier_circuit.rs:225
 * (`builder.build`) fixes and :253 (`data.prove`) runs four times per proof in plonky2 (constants/sigmas, wires,
 * zs/partial products, quotient; blowup 8, arity 16, 28 queries in `standard_recursion_config`, salted wires when zero
 * knowledge is on).  The gate constraints of the outer circuit are NOT part of this (upstream's circuit is not vendored). */
#define SIPP_SALT_SIZE 4
#define SIPP_FRI_MAX_ROUNDS 32
typedef struct {
    uint32_t rate_bits, cap_height, pow_bits, num_queries, pow_rule;
    uint32_t hiding;                           /* FriParams::hiding: salted oracles open their salt words too */
    uint32_t n_rounds;
    uint32_t arity_bits[SIPP_FRI_MAX_ROUNDS];  /* FriParams::reduction_arity_bits, each 1 .. 4 */
} sipp_fri_params;
/* FriReductionStrategy::ConstantArityBits(arity_bits, final_poly_bits) for polynomials of 2^degree_bits coefficients;
 * rate_bits and cap_height of *p must be set before the call */
void sipp_fri_const_arity(sipp_fri_params *p, uint32_t arity_bits, uint32_t final_poly_bits, uint32_t degree_bits);
/* a committed oracle, all pointers DEVICE memory owned by the caller:
 *   d_coeffs [n_polys][N] natural, d_lde [n_polys + n_salt][N << rate_bits] leaf order, d_tree (2 leaves x 4 u64, levels back to back) */
typedef struct {
    const uint64_t *d_coeffs;
    const uint64_t *d_lde;
    const uint64_t *d_tree;
    uint32_t n_polys, n_salt;
} sipp_oracle;
/* PolynomialBatch::from_values (from_coeffs != 0: from_coeffs) with explicit blowup / cap height and optional blinding:
 * d_salt = n_salt (0 or SIPP_SALT_SIZE) columns of N << rate_bits words in NATURAL LDE order (plonky2 draws them at random;
 * here the caller supplies them so that proofs are reproducible), appended to every leaf.  d_lde must hold ncols + n_salt
 * columns.  cap_out (host): 2^min(cap_height, log leaves) x 4 u64. */
int sipp_commit_batch_ex(sipp_ctx *ctx, const uint64_t *d_in, int from_coeffs, uint64_t *d_coeffs, uint64_t *d_lde,
                         uint64_t *d_tree, size_t ncols, uint32_t log_n, uint32_t rate_bits, uint32_t cap_height,
                         const uint64_t *d_salt, uint32_t n_salt, uint64_t *cap_out);
typedef struct { uint32_t oracle, col_begin, col_end; } sipp_poly_range;
typedef struct {
    uint64_t point[2];              /* extension element (c0, c1), not in the trace subgroup */
    uint32_t n_ranges;
    const sipp_poly_range *ranges;  /* FriBatchInfo::polynomials as ranges of oracle columns, in opening order */
} sipp_fri_batch;
/* plonky2 Challenger state as plain data: the caller's transcript goes in and comes out */
typedef struct {
    uint64_t state[12];
    uint64_t in_buf[8];
    uint64_t n_in;
    uint64_t out_buf[8];
    uint64_t n_out;
} sipp_challenger;
/* PolynomialBatch::prove_openings: the opened values are computed, written and observed batch by batch, then alpha, the
 * final polynomial, the commit phase, the proof of work and the query rounds.  Flat proof (u64 words):
 *   header[8]: "SIPPFRI1", n_rounds, final_len, num_queries, n_oracles, n_batches, total_len, log_n
 *   opened values (ext) per batch | commit caps | final_poly | pow_witness | query rounds
 * Supported range (SIPP_E_UNSUPPORTED outside it): log_n 10 .. 24, rate_bits 1 .. 3, arity_bits 1 .. 4 per round, cap_height
 * <= 8, and every committed layer holds at least 16 values; SIPP_E_BADARG for parameters the protocol itself forbids (a layer
 * with fewer leaves than its cap, arities that exceed the degree). */
size_t sipp_fri_proof_size(const sipp_oracle *oracles, size_t n_oracles, const sipp_fri_batch *batches, size_t n_batches,
                           uint32_t log_n, const sipp_fri_params *p);
int sipp_fri_prove_openings(sipp_ctx *ctx, const sipp_oracle *oracles, size_t n_oracles, const sipp_fri_batch *batches,
                            size_t n_batches, uint32_t log_n, const sipp_fri_params *p, sipp_challenger *ch,
                            uint64_t *proof_out, size_t proof_cap, size_t *proof_len);

/* ---- SURVEY.md section 8f, rank 2, continued: plonky2's WIRE PERMUTATION ARGUMENT -------------------------------------------------
 * The protocol-generic part of `data.prove(pw)` (reference src/verifier_circuit.rs:253) between the wires commitment and the opening
 * proof: plonk/prover.rs wires_permutation_partial_products_and_zs, the Z(1) = 1 and partial-product terms of
 * plonk/vanishing_poly.rs, compute_quotient_polys -- for any number of routed wires, chunk size (= quotient degree factor, a power
 * of two) and number of challenges, on the caller's wires and sigma polynomials.  The GATE constraints of the reference's circuit
 * are not part of this (the circuit is built by un-vendored crates). */
typedef struct {
    uint32_t num_routed_wires;   /* CircuitConfig::num_routed_wires (80 in standard_recursion_config) */
    uint32_t max_degree;         /* quotient_degree_factor = chunk size of the partial products (8); a power of two, 2 .. 64 */
    uint32_t num_challenges;     /* CircuitConfig::num_challenges (2); 1 .. 8 */
} sipp_plonk_params;
/* CommonCircuitData::num_partial_products = ceil(num_routed_wires / max_degree) - 1 */
uint32_t sipp_plonk_num_partial_products(const sipp_plonk_params *p);
/* d_wires, d_sigmas: [num_routed_wires][N] VALUES in natural row order (sigma[j][i] = k_col' w^row' of the wire that (j, i) maps to,
 * k_j = 7^j); betas, gammas: num_challenges host words.  d_out [num_challenges (1 + num_partial_products)][N] values in the column order
 * of the zs_partial_products commitment: Z_0 .. Z_{C-1}, then the partial products of challenge 0, of challenge 1, ... */
int sipp_plonk_zs_partial_products(sipp_ctx *ctx, const uint64_t *d_wires, const uint64_t *d_sigmas, uint32_t log_n,
                                   const sipp_plonk_params *p, const uint64_t *betas, const uint64_t *gammas, uint64_t *d_out);
/* compute_quotient_polys for the permutation terms, from the three committed oracles' LDEs (leaf order, blowup 2^rate_bits >=
 * max_degree; column strides N << rate_bits): d_chunks [num_challenges max_degree][N] receives the coefficient chunks (chunk d of
 * challenge c at row c max_degree + d) -- the input of sipp_commit_batch_ex(from_coeffs = 1). */
int sipp_plonk_quotient_chunks(sipp_ctx *ctx, const uint64_t *d_wires_lde, const uint64_t *d_sigmas_lde, const uint64_t *d_zs_lde,
                               uint32_t log_n, uint32_t rate_bits, const sipp_plonk_params *p, const uint64_t *betas,
                               const uint64_t *gammas, const uint64_t *alphas, uint64_t *d_chunks);
/* The same with the circuit's GATE-CONSTRAINT terms (round 4): plonk/vanishing_poly.rs builds vanishing_terms = vanishing_z_1_terms ++
 * vanishing_partial_products_terms ++ constraint_terms and runs ONE reduce_with_powers over all of them per challenge.  The gates of the
 * reference's circuit live in un-vendored crates, so their terms are the caller's: d_gate_terms [num_gate_terms][N << rate_bits], one
 * column per constraint, at the rows of the committed LDEs in leaf order (only the first N max_degree rows -- the quotient coset 7 <w_(N D)>
 * -- are read; any u64 congruent to the value).  A patched plonky2 whose gates stay on the CPU (or in its own kernels) evaluates them on
 * the wires LDE it committed through sipp_commit_batch_ex and hands the columns over.  num_gate_terms == 0 / NULL: the call above. */
int sipp_plonk_quotient_chunks_ex(sipp_ctx *ctx, const uint64_t *d_wires_lde, const uint64_t *d_sigmas_lde, const uint64_t *d_zs_lde,
                                  uint32_t log_n, uint32_t rate_bits, const sipp_plonk_params *p, const uint64_t *betas,
                                  const uint64_t *gammas, const uint64_t *alphas, const uint64_t *d_gate_terms, uint32_t num_gate_terms,
                                  uint64_t *d_chunks);
/* The whole argument as one call: commitments of sigmas, wires, zs_partial_products and quotient chunks, the transcript
 * (circuit_digest[4], public_inputs_hash[4], wires cap -> betas, gammas; zs cap -> alphas; quotient cap -> zeta) and one opening proof
 * (zeta: the four oracles; g zeta: the Z columns).  Flat proof (u64 words):
 *   header[8]: "SIPPPLK1", log_n, num_routed_wires, max_degree, num_challenges, total_len, 0, 0
 *   wires cap | zs_partial_products cap | quotient cap | sipp_fri_prove_openings' proof
 * The workspace must hold the four oracles (coefficients, LDE, tree each). */
size_t sipp_plonk_perm_proof_size(uint32_t log_n, const sipp_plonk_params *p, const sipp_fri_params *fp);
int sipp_plonk_perm_prove(sipp_ctx *ctx, const uint64_t *d_wires, const uint64_t *d_sigmas, uint32_t log_n, const sipp_plonk_params *p,
                          const sipp_fri_params *fp, const uint64_t circuit_digest[4], const uint64_t public_inputs_hash[4],
                          uint64_t *proof_out, size_t proof_cap, size_t *proof_len);

/* prove() of plonk/prover.rs except gate evaluation and witness generation: as sipp_plonk_perm_prove, plus
 *   - the gate-constraint terms in the quotient (d_gate_terms as for sipp_plonk_quotient_chunks_ex; may be NULL / 0),
 *   - the PUBLIC INPUTS: public_inputs_hash = hash_n_to_hash_no_pad(public_inputs) is computed here and observed after circuit_digest;
 *     the inputs themselves travel behind the opening proof,
 *   - oracles the caller committed already with sipp_commit_batch_ex at this blowup and cap height (both optional): sigmas_oracle (the
 *     constants_sigmas commitment is made once per circuit), and wires_oracle + wires_cap (the caller needed the wires LDE for its gates:
 *     the same buffers are opened here, nothing is transformed twice).  d_wires / d_sigmas VALUES are always needed (Z and the partial
 *     products are computed from values).
 * Flat proof "SIPPPLK2": header[8] = magic, log_n, num_routed_wires, max_degree, num_challenges, total_len, num_gate_terms,
 * n_public_inputs | wires cap | zs_partial_products cap | quotient cap | opening proof | public_inputs.
 * proof_cap >= sipp_plonk_perm_proof_size(..) + n_public_inputs. */
int sipp_plonk_prove_ex(sipp_ctx *ctx, const uint64_t *d_wires, const uint64_t *d_sigmas, const sipp_oracle *wires_oracle,
                        const uint64_t *wires_cap, const sipp_oracle *sigmas_oracle, uint32_t log_n, const sipp_plonk_params *p,
                        const sipp_fri_params *fp, const uint64_t circuit_digest[4], const uint64_t *public_inputs,
                        uint32_t n_public_inputs, const uint64_t *d_gate_terms, uint32_t num_gate_terms, uint64_t *proof_out,
                        size_t proof_cap, size_t *proof_len);

/* ---- round 5: GATES AS DATA -- evaluate_gate_constraints on the device ---------------------------------------------------------------
 * plonk/vanishing_poly.rs evaluate_gate_constraints / gates/gate.rs eval_filtered / gates/selectors.rs for a gate set the CALLER describes
 * (the reference's circuit, src/verifier_circuit.rs:213-226, is built by un-vendored crates): a gate = its selector (which selector
 * column, which value = the gate's index, the range of gate indices of its selector group) and a PROGRAM -- one polynomial per
 * constraint over the row's wires, the constant columns and the public-inputs hash, as a sum of monomials.  Constraint term j of the
 * circuit = sum over gates of filter_g * constraint_{g,j};  filter_g(s) = prod_{i in group, i != g} (i - s) * (UNUSED - s when the
 * circuit has more than one selector column), UNUSED = 2^32 - 1.  The terms follow the permutation terms in ONE reduce_with_powers.
 * Program words (int64, host memory): per constraint  n_mono, then per monomial  coef, n_factors, (kind, index) x n_factors;
 * kind 0 = wire, 1 = constant column (index into the num_constants columns, selectors included), 2 = public_inputs_hash word. */
typedef struct {
    uint32_t selector_index, row, group_lo, group_hi;
    uint32_t prog_offset, num_constraints;
} sipp_plonk_gate;
typedef struct {
    uint32_t num_wires;       /* all wire columns of the wires oracle (>= num_routed_wires; 135 / 136 in plonky2's standard configs) */
    uint32_t num_constants;   /* columns of the constants_sigmas oracle in front of the sigmas: num_selectors selector columns, then gate constants */
    uint32_t num_selectors;
    uint32_t num_gates;
    const sipp_plonk_gate *gates;
    const int64_t *programs;
    uint32_t program_words;
} sipp_plonk_circuit;
/* prove() of plonk/prover.rs except witness generation: d_wires [num_wires][N] (the first num_routed_wires are routed) and
 * d_constants_sigmas [num_constants + num_routed_wires][N] are VALUES in natural row order; the gate constraints are evaluated on the
 * quotient coset by the interpreter inside the quotient kernel (no column per constraint is materialised).  Optional pre-committed
 * oracles as for sipp_plonk_prove_ex (constants_sigmas once per circuit).  Flat proof "SIPPPLK3": header[16] = magic, log_n,
 * num_routed_wires, max_degree, num_challenges, total_len, num_wires, num_constants, num_selectors, num_gates, num_gate_constraints,
 * n_public_inputs, 0, 0, 0, 0 | wires cap | zs_partial_products cap | quotient cap | opening proof (zeta: every column of the four
 * oracles constants_sigmas, wires, zs_partial_products, quotient chunks; g zeta: the Z columns) | public_inputs.
 * SIPP_E_BADARG for a malformed circuit (an operand out of range, a program that runs past program_words). */
size_t sipp_plonk_gates_proof_size(uint32_t log_n, const sipp_plonk_params *p, const sipp_fri_params *fp, const sipp_plonk_circuit *c,
                                   uint32_t n_public_inputs);
int sipp_plonk_prove_gates(sipp_ctx *ctx, const uint64_t *d_wires, const uint64_t *d_constants_sigmas, const sipp_oracle *wires_oracle,
                           const uint64_t *wires_cap, const sipp_oracle *constants_sigmas_oracle, uint32_t log_n, const sipp_plonk_params *p,
                           const sipp_fri_params *fp, const sipp_plonk_circuit *c, const uint64_t circuit_digest[4],
                           const uint64_t *public_inputs, uint32_t n_public_inputs, uint64_t *proof_out, size_t proof_cap, size_t *proof_len);

/* ---- round 6: the gates' WITNESS GENERATORS on the device ---------------------------------------------------------------------------
 * plonky2's prove() starts with generate_partial_witness (iop/generator.rs; reference src/verifier_circuit.rs:253 `data.prove(pw)`): every
 * gate instance's SimpleGenerator fills the wires its constraints determine from the wires the caller (or a copy constraint) set.  Here the
 * generators of the gate families a recursive verifier is made of run for ALL rows of their gate at once, one lane per row, in place on
 * the wire table the prover reads next: a generator = (family, the selector column and value that mark its rows, layout parameters p[]).
 *   SIPP_GEN_ARITHMETIC     p = n_ops, const col c0, const col c1       w[4k+3] = c0 w[4k] w[4k+1] + c1 w[4k+2]          (ArithmeticBaseGenerator)
 *   SIPP_GEN_BASE_SPLIT     p = n_limbs, bits per limb                  w[1+i] = limb i of the integer w[0]                (BaseSplitGenerator)
 *   SIPP_GEN_CONSTANT       p = n, first const col                      w[i] = const[p1 + i]                               (ConstantGate)
 *   SIPP_GEN_PUBLIC_INPUT   --                                          w[i] = public_inputs_hash[i], i < 4                (PublicInputGate)
 *   SIPP_GEN_U32_MUL_ADD    p = n_ops, stride, limbs per half           per op at b: (lo, hi) of w[b] w[b+1] + w[b+2] into b+3, b+4, then the
 *                                                                       2-bit limbs of lo and of hi                        (U32ArithmeticGenerator)
 *   SIPP_GEN_RANDOM_ACCESS  p = copies, stride, bits                    per copy at b: index, claimed, 2^bits items, bits: claimed = items[index],
 *                                                                       bits of index                                      (RandomAccessGenerator)
 *   SIPP_GEN_REDUCING       p = K, W                                    alpha (2), old acc (2), K coefficients, K accumulators (2 each) over
 *                                                                       F[X]/(X^2 - W): acc_i = acc_(i-1) alpha + c_i      (ReducingGenerator)
 *   SIPP_GEN_POSEIDON       p = in, out, sbox                           the permutation of w[in .. in+12): every S-box input of rounds 1 .. 29
 *                                                                       (36 + 22 + 48 wires from sbox) and the 12 outputs  (PoseidonGenerator)
 * The generators are ROW-LOCAL: values that reach a gate's inputs through copy constraints from another gate's outputs have to be there
 * already (the caller orders its calls by level); d_constants = the circuit's constant columns [num_constants][N] (selectors first: the
 * front of d_constants_sigmas).  SIPP_E_BADARG for a layout that leaves the wire table or an unknown family. */
#define SIPP_GEN_ARITHMETIC 1
#define SIPP_GEN_BASE_SPLIT 2
#define SIPP_GEN_CONSTANT 3
#define SIPP_GEN_PUBLIC_INPUT 4
#define SIPP_GEN_U32_MUL_ADD 5
#define SIPP_GEN_RANDOM_ACCESS 6
#define SIPP_GEN_REDUCING 7
#define SIPP_GEN_POSEIDON 8
typedef struct {
    uint32_t kind, selector_index, row;
    uint32_t p[5];
} sipp_plonk_generator;
int sipp_plonk_generate_witness(sipp_ctx *ctx, uint64_t *d_wires, const uint64_t *d_constants, uint32_t log_n, uint32_t num_wires,
                                uint32_t num_constants, const sipp_plonk_generator *gens, size_t n_gens, const uint64_t public_inputs_hash[4]);
/* The same for a circuit whose COPY CONSTRAINTS carry outputs of one row to inputs of another (hash chains, Merkle paths, accumulators):
 * generate_partial_witness resolves that order with a work list at proving time; here the circuit builder fixes it once as a SCHEDULE --
 * the rows sorted by level and, per level, the cells its outputs feed (cell = wire * N + row).  For l = 0 .. n_levels - 1: the generators
 * on d_rows[level_offsets[l] .. level_offsets[l + 1]) (one lane per row; a row runs the generator whose selector value it holds), then
 * d_wires[d_copy_dst[k]] = d_wires[d_copy_src[k]] for k in [copy_offsets[l], copy_offsets[l + 1]).  Two small launches per level: the
 * sequence is captured once as a hipGraph per ctx and replayed while the arguments stay the same (SIPP_ROUTE_WITNESS_NO_GRAPH, or an
 * enabled profile, launches them one by one).  d_rows / d_copy_* are device arrays, the offsets host arrays of n_levels + 1 entries.
 * SIPP_E_BADARG also for a row or cell outside the table (checked on the device, reported after the last level). */
typedef struct {
    uint32_t n_levels;
    const uint32_t *d_rows;
    const uint32_t *level_offsets;
    const uint64_t *d_copy_src;
    const uint64_t *d_copy_dst;
    const uint32_t *copy_offsets;
} sipp_plonk_schedule;
int sipp_plonk_generate_witness_levels(sipp_ctx *ctx, uint64_t *d_wires, const uint64_t *d_constants, uint32_t log_n, uint32_t num_wires,
                                       uint32_t num_constants, const sipp_plonk_generator *gens, size_t n_gens,
                                       const uint64_t public_inputs_hash[4], const sipp_plonk_schedule *sched);

/* ---- round 6: CIRCUIT DATA -- `builder.build()` / `data.prove(pw)` / `data.verify(proof)` with HOST pointers only -----------------------
 * The reference builds its circuit once (src/verifier_circuit.rs:225 `builder.build::<C>()`), then proves and verifies through the
 * CircuitData (:253 `data.prove(pw)`, :254 `data.verify(proof)`).  The entry points above take DEVICE buffers of the caller; a caller
 * without HIP bindings of its own (the Rust shim) uses these instead: everything the circuit fixes -- the gate set, the generators
 * with their level schedule, the constants_sigmas VALUES -- goes in once and stays on the device with its commitment; a proof takes the
 * wire table with the INPUT cells set (plonky2's PartialWitness; every generated cell is overwritten) and the public inputs from host
 * memory and returns the flat "SIPPPLK3" proof to host memory.  One proof at a time per circuit data; it belongs to its ctx (stream,
 * arena >= sipp_circuit_workspace_bytes) and must be destroyed before it.
 * circuit_digest: 4 words, or NULL = hash_no_pad(constants_sigmas cap || log_n, num_wires, num_routed_wires, num_constants,
 * num_selectors, num_gates) (plonky2 hashes the cap and its common data the same way; the exact word list is this library's).
 * sched may be NULL (row-local generators: sipp_plonk_generate_witness); its arrays are HOST arrays (copied). */
typedef struct sipp_circuit_data sipp_circuit_data;
typedef struct {
    uint32_t n_levels;
    const uint32_t *rows;
    const uint32_t *level_offsets;
    const uint64_t *copy_src;
    const uint64_t *copy_dst;
    const uint32_t *copy_offsets;
} sipp_plonk_schedule_host;
size_t sipp_circuit_workspace_bytes(uint32_t log_n, const sipp_plonk_params *p, const sipp_fri_params *fp, const sipp_plonk_circuit *c);
int sipp_circuit_build(sipp_ctx *ctx, uint32_t log_n, const sipp_plonk_params *p, const sipp_fri_params *fp, const sipp_plonk_circuit *c,
                       const uint64_t *constants_sigmas, const sipp_plonk_generator *gens, size_t n_gens,
                       const sipp_plonk_schedule_host *sched, const uint64_t *circuit_digest, sipp_circuit_data **out);
void sipp_circuit_destroy(sipp_circuit_data *cd);
/* the verifier's half (VerifierOnlyCircuitData): cap_out 2^cap_height x 4 words, digest_out 4 words */
int sipp_circuit_verifier_data(const sipp_circuit_data *cd, uint64_t *cap_out, uint64_t *digest_out);
size_t sipp_circuit_proof_size(const sipp_circuit_data *cd, uint32_t n_public_inputs);
/* wires: HOST [num_wires][N]; the proof's public inputs are bound through public_inputs_hash (PublicInput generator / gate) */
int sipp_circuit_prove(sipp_circuit_data *cd, const uint64_t *wires, const uint64_t *public_inputs, uint32_t n_public_inputs,
                       uint64_t *proof_out, size_t proof_cap, size_t *proof_len);
int sipp_circuit_verify(const sipp_circuit_data *cd, const uint64_t *proof, size_t len, int *reason);

/* ---- verification of the generic proofs (host code like sipp_stark_verify; stages in *reason, may be NULL) -------------------------
 * sipp_fri_verify_openings: PolynomialBatch::verify_openings over a proof of sipp_fri_prove_openings -- caps[o] = the cap of oracle o
 * (2^cap_height x 4 words), ncols / n_salt (may be NULL = 0) per oracle, the batches as proved; the caller's transcript goes in and
 * comes out (the opened values are observed as they are read).  Stages: 100 header, 106 truncated openings, 120 .. 133 as in
 * sipp_stark_verify, 140 trailing words, 141 a non-canonical word.
 * sipp_plonk_verify_gates: plonk/verifier.rs for a proof of sipp_plonk_prove_gates ("SIPPPLK3") -- what `data.verify(proof)` (reference
 * src/verifier_circuit.rs:254) does with the outer proof: the gate constraints at zeta from the OPENED constants and wires through the
 * same gate programs, Z(1) = 1, the partial products, one reduce_with_powers per challenge against the quotient chunks, the opening
 * proof.  constants_sigmas_cap and circuit_digest are the circuit's (verifier data).  Stages: 201 header / circuit / parameters,
 * 202 size, 203 truncated, 210 the vanishing polynomial does not meet the quotient, then the FRI stages above. */
int sipp_fri_verify_openings(const uint64_t *proof, size_t len, const uint64_t *const *caps, const uint32_t *ncols, const uint32_t *n_salt,
                             size_t n_oracles, const sipp_fri_batch *batches, size_t n_batches, uint32_t log_n, const sipp_fri_params *p,
                             sipp_challenger *ch, int *reason);
int sipp_plonk_verify_gates(const uint64_t *proof, size_t len, const uint64_t *constants_sigmas_cap, const sipp_plonk_params *p,
                            const sipp_fri_params *fp, const sipp_plonk_circuit *c, const uint64_t circuit_digest[4], int *reason);

/* ---- building blocks (device buffers; used by the parity tests and bench.py) -- */
/* plonky2 fft()/ifft(): natural order in, natural order out, in place.
 * d_cols is [ncols][col_stride] u64 with the first 2^log_n entries of each column used. */
int sipp_ntt_batch(sipp_ctx *ctx, uint64_t *d_cols, size_t col_stride, size_t ncols, uint32_t log_n, int inverse);
/* PolynomialBatch::from_values, column-major and transpose-free:
 *   d_values [ncols][N] natural order  ->  d_coeffs [ncols][N] (natural, may alias d_values)
 *   d_lde [ncols][N << rate_bits] in LEAF order (position j = natural LDE row bitrev(j)).
 * ALIGNMENT (this call, sipp_commit_batch, sipp_commit_batch_ex and the oracles handed to the plonk entry points): for columns of
 * 2^13 rows and more every device pointer must be 16-byte aligned (the transforms move 16 bytes per lane); an 8-byte-aligned
 * view is SIPP_E_BADARG.  hipMalloc / torch allocations are; slices at odd u64 offsets are not. */
int sipp_lde_batch(sipp_ctx *ctx, const uint64_t *d_values, uint64_t *d_coeffs, uint64_t *d_lde, size_t ncols,
                   uint32_t log_n);
/* hash_or_noop of every leaf: d_lde [ncols][n_leaves] leaf order -> d_digests [n_leaves][4] */
int sipp_poseidon_leaves(sipp_ctx *ctx, const uint64_t *d_lde, size_t ncols, uint32_t log_leaves,
                         uint64_t *d_digests);
/* Merkle levels above the leaf digests; d_tree holds every level back to back
 * (level l at digest offset sum_{i<l} n_leaves >> i), level 0 = leaf digests (input).
 * cap_out (host, 2^cap_height * 4 u64) receives the cap. */
int sipp_merkle_cap(sipp_ctx *ctx, uint64_t *d_tree, uint32_t log_leaves, uint64_t *cap_out);
/* values -> (coeffs, lde, tree, cap) in one call; d_tree sized 2 * n_leaves * 4 u64 */
int sipp_commit_batch(sipp_ctx *ctx, const uint64_t *d_values, uint64_t *d_coeffs, uint64_t *d_lde,
                      uint64_t *d_tree, size_t ncols, uint32_t log_n, uint64_t *cap_out);
/* Trace generation only (test surface for the trace-fill kernels): ios as for sipp_*_exp_prove;
 * d_trace receives the column-major trace [width][2^log_rows] in natural row order, where
 * width = main_cols of sipp_stark_shape.  d_trace must hold width * 2^log_rows u64. */
int sipp_trace_build(sipp_ctx *ctx, int kind, const uint32_t *ios, size_t num_io, uint64_t *d_trace);
/* batched Poseidon permutation of n states, d_states [n][12] in place (KAT surface) */
int sipp_poseidon_permute(sipp_ctx *ctx, uint64_t *d_states, size_t n);
/* the HOST permutation of the Fiat-Shamir challenger (no GPU involved): n states [n][12] in host memory, in place.
 * impl: -1 = the implementation the provers use on this CPU, 0 = portable scalar, 1 = scalar with look-ahead partial
 * rounds, 2 / 3 = AVX-512 (SIPP_E_UNSUPPORTED when the CPU lacks it).  All are bit-identical. */
int sipp_host_poseidon_permute(uint64_t *states, size_t n, int impl);

/* ---- measurement ------------------------------------------------------------- */
/* Per-kernel HIP-event timing on the ctx stream (the stream the kernels run on).
 * enable != 0 starts bracketing every launch with an event pair. */
int sipp_profile_enable(sipp_ctx *ctx, int enable);
int sipp_profile_reset(sipp_ctx *ctx);
/* writes a JSON object {"kernel": {"calls": n, "ms": total}, ...} into buf */
int sipp_profile_report(sipp_ctx *ctx, char *buf, size_t cap);
/* wall-clock bracket on the ctx stream */
int sipp_timer_start(sipp_ctx *ctx);
int sipp_timer_stop(sipp_ctx *ctx, float *ms);

#ifdef __cplusplus
}
#endif
#endif
