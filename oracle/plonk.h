/*
 * oracle/plonk.h -- CPU restatement of the WIRE PERMUTATION ARGUMENT of plonky2's outer prover: what `data.prove(pw)` at reference
 * src/verifier_circuit.rs:253 runs between the wires commitment and the opening proof -- plonk/prover.rs
 * `wires_permutation_partial_products_and_zs`, plonk/vanishing_poly.rs `eval_vanishing_poly_base_batch` (its Z(1) = 1 and
 * partial-product terms), `compute_quotient_polys`, plonk/plonk_common.rs (`quotient_chunk_products`, `partial_products_and_z_gx`,
 * `check_partial_products`, `reduce_with_powers`), plonk/verifier.rs' check of the quotient at zeta.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED: plonky2 @ InternetMaximalism/plonky2 541e127 is not vendored (reference Cargo.toml:21), the
 * sequence is restated from its published structure; a second, independent reading lives in oracle/py/plonky2_generic.py.
 * The GATE constraints of the reference's circuit are not part of this (the circuit is built by un-vendored crates): wires and sigmas
 * are the caller's data, any `num_routed_wires`, chunk size (= quotient degree factor, a power of two) and `num_challenges`.
 */
#ifndef ORACLE_PLONK_H
#define ORACLE_PLONK_H
#include "fri.h"

typedef struct {
    uint32_t num_routed_wires;   /* CircuitConfig::num_routed_wires (80 in standard_recursion_config) */
    uint32_t max_degree;         /* quotient_degree_factor = chunk size of the partial products (8); a power of two >= 2 */
    uint32_t num_challenges;     /* CircuitConfig::num_challenges (2) */
} orc_plonk_params;

static inline uint32_t orc_plonk_num_prods(const orc_plonk_params *p) {   /* CommonCircuitData::num_partial_products */
    return (p->num_routed_wires + p->max_degree - 1) / p->max_degree - 1;
}
static inline uint32_t orc_plonk_zs_cols(const orc_plonk_params *p) { return p->num_challenges * (1 + orc_plonk_num_prods(p)); }

/* k_is = get_unique_coset_shifts: 7^j */
uint64_t orc_plonk_k_i(uint32_t j);
/* sigma polynomial VALUES for a wire permutation: sigma[j][i] = k_{col'} w^{row'} where (col', row') = perm[j * N + i] (index col' * N + row') */
void orc_plonk_sigmas_from_perm(const uint32_t *perm, uint32_t num_routed, unsigned log_n, uint64_t *sigmas);

/* wires [R][N], sigmas [R][N] values in natural order -> out [C (1 + num_prods)][N] values: Z_0 .. Z_{C-1}, then the partial products of
 * challenge 0, of challenge 1, ... (the column order of the zs_partial_products commitment) */
void orc_plonk_zs_partial_products(const uint64_t *wires, const uint64_t *sigmas, unsigned log_n, const orc_plonk_params *p,
                                   const uint64_t *betas, const uint64_t *gammas, uint64_t *out);

/* compute_quotient_polys for the permutation terms: coefficient arrays in, [C max_degree][N] coefficient chunks out (chunk d of
 * challenge c at row c max_degree + d).  Returns -1 if a quotient does not fit max_degree N coefficients (constraints violated). */
int orc_plonk_quotient_chunks(const uint64_t *wires_c, const uint64_t *sigmas_c, const uint64_t *zs_c, unsigned log_n,
                              const orc_plonk_params *p, const uint64_t *betas, const uint64_t *gammas, const uint64_t *alphas,
                              uint64_t *out);

/* ---- the same with the circuit's GATE-CONSTRAINT terms (round 4): vanishing_terms = z_1 terms ++ partial-product terms ++ constraint
 * terms, one reduce_with_powers over all of them (plonk/vanishing_poly.rs).  The reference's circuit is not vendored, so the terms are
 * the caller's; the test circuit below stands in for a gate set: num_mul product gates, term k = w_{3k} w_{3k+1} - w_{3k+2} on every row. */
typedef struct {
    uint32_t num_mul;
} orc_plonk_gates;
/* that circuit's terms on the quotient coset 7 <w_{N D}>, natural order: out [num_mul][N D] */
void orc_plonk_gate_terms_coset(const uint64_t *wires_c, unsigned log_n, unsigned log_d, const orc_plonk_gates *g, uint64_t *out);
int orc_plonk_quotient_chunks_ex(const uint64_t *wires_c, const uint64_t *sigmas_c, const uint64_t *zs_c, unsigned log_n,
                                 const orc_plonk_params *p, const uint64_t *betas, const uint64_t *gammas, const uint64_t *alphas,
                                 const uint64_t *gate_terms, uint32_t n_gate_terms, uint64_t *out);
void orc_plonk_eval_vanishing_ex(gl2 x, const gl2 *wires, const gl2 *sigmas, const gl2 *zs, const gl2 *zs_next, const gl2 *pps,
                                 unsigned log_n, const orc_plonk_params *p, const uint64_t *betas, const uint64_t *gammas,
                                 const uint64_t *alphas, const gl2 *gate_terms, uint32_t n_gate_terms, gl2 *out);
/* the flow with gates and public inputs: flat proof "SIPPPLK2" = header[8] (.., n_gate_terms, n_public_inputs) | the three caps | opening
 * proof | public_inputs; transcript: circuit_digest, hash_no_pad(public_inputs), wires cap -> ... as above */
int orc_plonk_prove_ex(const uint64_t *wires, const uint64_t *sigmas, unsigned log_n, const orc_plonk_params *p, const orc_fri_params *fp,
                       const uint64_t circuit_digest[4], const uint64_t *public_inputs, uint32_t n_public_inputs, const orc_plonk_gates *g,
                       uint64_t **proof, size_t *len);
int orc_plonk_verify_ex(const uint64_t *proof, size_t len, const uint64_t *sigmas_cap, const orc_plonk_params *p, const orc_fri_params *fp,
                        const uint64_t circuit_digest[4], const orc_plonk_gates *g);

/* ---- round 5: GATES AS DATA (plonk_gates.c) ------------------------------------------------------------------------------------------
 * plonky2's evaluate_gate_constraints (plonk/vanishing_poly.rs) with the gate set given as DATA: a gate = its selector (which selector
 * column, which value = the gate's index, the range of gate indices its group covers: gates/selectors.rs) and a PROGRAM -- one polynomial
 * per constraint over the row's wires, constant columns and the public-inputs hash, as a sum of monomials.  Constraint term j of the
 * circuit = sum over gates of filter_g * constraint_{g,j} (gate.rs eval_filtered adds into a shared vector), filter_g(s) = prod_{i in group,
 * i != g} (i - s) * (UNUSED - s if the circuit has more than one selector column), UNUSED = 2^32 - 1.
 * Oracles: constants_sigmas = [selectors | gate constants | sigmas of the routed wires] (num_constants columns in front of the sigmas),
 * wires = ALL num_wires columns (the first num_routed_wires are routed), zs_partial_products, quotient chunks.
 * Program words (int64): per constraint  n_mono, then per monomial  coef, n_factors, (kind, index) x n_factors;  kind 0 = wire,
 * 1 = constant column (index into the num_constants columns: selectors included), 2 = public_inputs_hash word. */
typedef struct {
    uint32_t selector_index, row, group_lo, group_hi;
    uint32_t prog_offset, num_constraints;
} orc_plonk_gate;
typedef struct {
    uint32_t num_wires, num_constants, num_selectors, num_gates;
    const orc_plonk_gate *gates;
    const int64_t *programs;
    uint32_t program_words;
} orc_plonk_circuit;
#define ORC_UNUSED_SELECTOR 0xffffffffULL
uint32_t orc_plonk_num_gate_constraints(const orc_plonk_circuit *c);
/* 0 if the programs are well formed (operands in range, lengths consistent) */
int orc_plonk_circuit_check(const orc_plonk_circuit *c, const orc_plonk_params *p);
/* the circuit's constraint terms at ONE row / point over the base field: wires[num_wires], consts[num_constants], pih[4] -> out[num_gate_constraints] */
void orc_plonk_gate_constraints_base(const orc_plonk_circuit *c, const uint64_t *wires, const uint64_t *consts, const uint64_t pih[4], uint64_t *out);
void orc_plonk_gate_constraints_ext(const orc_plonk_circuit *c, const gl2 *wires, const gl2 *consts, const uint64_t pih[4], gl2 *out);
/* flat proof "SIPPPLK3": header[16] = magic, log_n, num_routed_wires, max_degree, num_challenges, total_len, num_wires, num_constants,
 * num_selectors, num_gates, num_gate_constraints, n_public_inputs, 0, 0, 0, 0 | wires cap | zs_partial_products cap | quotient cap |
 * opening proof (zeta: constants_sigmas, wires, zs_partial_products, quotient chunks -- all columns; g zeta: the Z columns) | public_inputs.
 * wires [num_wires][N], constants_sigmas [num_constants + num_routed_wires][N]: VALUES, natural row order. */
int orc_plonk_prove_gates(const uint64_t *wires, const uint64_t *constants_sigmas, unsigned log_n, const orc_plonk_params *p,
                          const orc_fri_params *fp, const orc_plonk_circuit *c, const uint64_t circuit_digest[4], const uint64_t *public_inputs,
                          uint32_t n_public_inputs, uint64_t **proof, size_t *len);
/* cs_cap: the verifier's copy of the constants_sigmas commitment */
int orc_plonk_verify_gates(const uint64_t *proof, size_t len, const uint64_t *cs_cap, const orc_plonk_params *p, const orc_fri_params *fp,
                           const orc_plonk_circuit *c, const uint64_t circuit_digest[4]);

/* the verifier's side: vanishing terms at an extension point from opened values, reduced with the powers of every alpha */
void orc_plonk_eval_vanishing(gl2 x, const gl2 *wires, const gl2 *sigmas, const gl2 *zs, const gl2 *zs_next, const gl2 *pps,
                              unsigned log_n, const orc_plonk_params *p, const uint64_t *betas, const uint64_t *gammas,
                              const uint64_t *alphas, gl2 *out);

/* the whole flow as a flat proof:
 *   header[8]: "SIPPPLK1", log_n, num_routed_wires, max_degree, num_challenges, total_len, 0, 0
 *   wires cap | zs_partial_products cap | quotient cap | opening proof (orc_fri_prove_openings: zeta batch over the four oracles
 *   sigmas, wires, zs_partial_products, quotient chunks; g zeta batch over the Z columns)
 * Transcript: circuit_digest[4], public_inputs_hash[4], wires cap -> betas, gammas; zs cap -> alphas; quotient cap -> zeta. */
int orc_plonk_perm_prove(const uint64_t *wires, const uint64_t *sigmas, unsigned log_n, const orc_plonk_params *p,
                         const orc_fri_params *fp, const uint64_t circuit_digest[4], const uint64_t public_inputs_hash[4],
                         uint64_t **proof, size_t *len);
/* sigmas_cap: the verifier's copy of the constants_sigmas commitment (part of the verifier data in plonky2) */
int orc_plonk_perm_verify(const uint64_t *proof, size_t len, const uint64_t *sigmas_cap, const orc_plonk_params *p,
                          const orc_fri_params *fp, const uint64_t circuit_digest[4], const uint64_t public_inputs_hash[4]);
/* ---- round 6: the gates' WITNESS GENERATORS (plonk_witness.c) -----------------------------------------------------------------------
 * plonky2's generate_partial_witness (iop/generator.rs), first step of prove(): the SimpleGenerator of every gate instance fills the wires
 * the gate's constraints determine.  Row-local, family by family, layouts as data (the product's sipp_plonk_generator, include/sipp_hip.h):
 * kind 1 arithmetic (p: n_ops, c0 column, c1 column), 2 base split (n_limbs, bits), 3 constant (n, first column), 4 public input,
 * 5 U32 multiply-add (n_ops, stride, 2-bit limbs per half), 6 random access (copies, stride, bits), 7 reducing (K, W), 8 Poseidon (in, out,
 * first S-box wire).  wires [num_wires][N] in place, consts [num_constants][N]; rows of a generator = consts[selector_index][row] == row value.
 * Returns 0, or -1 for a layout that leaves the tables / an unknown family. */
typedef struct {
    uint32_t kind, selector_index, row;
    uint32_t p[5];
} orc_plonk_generator;
int orc_plonk_generate_witness(uint64_t *wires, const uint64_t *consts, unsigned log_n, uint32_t num_wires, uint32_t num_constants,
                               const orc_plonk_generator *gens, size_t n_gens, const uint64_t pih[4]);
/* the same LEVEL BY LEVEL for a circuit whose copy constraints carry outputs to inputs of other rows (generate_partial_witness' work list,
 * fixed at circuit-build time): for l = 0 .. n_levels - 1 run the generators on rows[level_offsets[l] .. level_offsets[l + 1]), then copy
 * cell copy_src[k] to cell copy_dst[k] (cell = wire * N + row) for k in [copy_offsets[l], copy_offsets[l + 1]).  -1 also for a row or cell
 * outside the table. */
int orc_plonk_generate_witness_levels(uint64_t *wires, const uint64_t *consts, unsigned log_n, uint32_t num_wires, uint32_t num_constants,
                                      const orc_plonk_generator *gens, size_t n_gens, const uint64_t pih[4], uint32_t n_levels,
                                      const uint32_t *rows, const uint32_t *level_offsets, const uint64_t *copy_src, const uint64_t *copy_dst,
                                      const uint32_t *copy_offsets);
#endif
