"""The reference's BLS example end to end with every heavy step on the GPU (src/bin/bls_aggregation.rs:88-160 `main`, natively, plus
what `verify_bls_aggregation` (:40-85) delegates to STARKs):

    keys, messages  ->  ms = map_to_g2(messages)                       sipp_map_to_g2            (:100-104)
                    ->  signatures, aggregated signature               (host big integers: the signer's side)
                    ->  a = pks + [-G1], b = ms + [agg]                (:111-115)
                    ->  inner_product(a, b) == 1                       sipp_inner_product        (:117)
                    ->  sipp_prove_native / sipp_verify_native         (:118-119)
                    ->  pairing(final_A, final_B) == final_Z           sipp_inner_product        (:120)
                    ->  batch_map_to_g2_circuit                        sipp_map_to_g2_prove + the cofactor G2ExpStark proof  (:65)
                    ->  sipp_verifier_circuit's three STARKs           sipp_instance_prove       (:73)
                    ->  pairing_circuit(final_A, final_B) == final_Z   sipp_pairing_prove        (:76-77)

Every proof goes through the oracle's verifier; the outer plonky2 proof is not built."""
import os
import sys

import numpy as np
import pytest

from tests import _oracle, _verify

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle", "py"))

pytestmark = pytest.mark.gpu


def test_bls_aggregation_n8():
    import random
    import bn254 as bn
    import sipp_amd
    n = 8
    rnd = random.Random(0xb15)
    sks = [rnd.randrange(1, bn.R) for _ in range(n - 1)]
    pks = [bn.g1_mul(bn.G1, sk) for sk in sks]
    msgs = [(rnd.randrange(bn.P), rnd.randrange(bn.P)) for _ in range(n - 1)]
    words = np.array([bn.fq_to_u32(u[0]) + bn.fq_to_u32(u[1]) for u in msgs], dtype=np.uint32)
    L = sipp_amd.lib()
    ctx = sipp_amd.Ctx(workspace_bytes=max(1 << 30, L.sipp_workspace_bytes(1, 2 * (n - 1)), L.sipp_workspace_bytes(3, n - 1), L.sipp_workspace_bytes(6, 1)))
    try:
        map_recs, cof_recs, ms_words = ctx.map_to_g2(words)
        ms = [((bn.u32_to_fq(list(w[0:8])), bn.u32_to_fq(list(w[8:16]))), (bn.u32_to_fq(list(w[16:24])), bn.u32_to_fq(list(w[24:32]))))
              for w in ms_words]
        assert all(bn.g2_on_curve(m) and bn.g2_mul(m, bn.R) is None for m in ms)          # in G2
        sigs = [bn.g2_mul(m, sk) for m, sk in zip(ms, sks)]
        agg = None
        for s in sigs:
            agg = bn.g2_add(agg, s)
        a = pks + [bn.g1_neg(bn.G1)]
        b = ms + [agg]
        A = np.array([bn.g1_to_u32(p) for p in a], dtype=np.uint32)
        B = np.array([bn.g2_to_u32(q) for q in b], dtype=np.uint32)
        one = np.zeros(96, dtype=np.uint32)
        one[0] = 1
        assert (ctx.inner_products(A, B)[0] == one).all()                                  # the aggregated signature verifies
        proof = ctx.prove_native(A, B)
        ok, st, ios = ctx.verify_native(A, B, proof)
        assert ok
        # statement layout (src/statements.rs:24-39): ... final_A (16) | final_B (32) | final_Z (96) at the end
        fa, fb, fz = st[-144:-128], st[-128:-96], st[-96:]
        assert (ctx.inner_products(fa.reshape(1, 16), fb.reshape(1, 32))[0] == fz).all()  # pairing(final_A, final_B) == final_Z
        # a forged aggregate (one signature missing) does not verify
        Bbad = B.copy()
        Bbad[-1] = bn.g2_to_u32(bn.g2_add(agg, bn.g2_neg(sigs[0])))
        assert not (ctx.inner_products(A, Bbad)[0] == one).all()
        # the STARKs: messages -> points, the cofactor clearing, and the three of the SIPP verifier
        pf_map = ctx.prove(3, map_recs)
        pf_cof = ctx.prove(1, cof_recs)
        # the in-circuit final pairing (:76-77) as a STARK obligation: the record (final_A, final_B, final_Z)
        final_rec = np.concatenate([fa, fb, fz]).reshape(1, 144)
        pf_pair = ctx.prove(6, final_rec)
        with pytest.raises(sipp_amd.SippError):          # another final_Z has no proof
            bad = final_rec.copy()
            bad[0, 60] ^= 4
            ctx.prove(6, bad)
    finally:
        ctx.close()
    assert _verify.both_accept(pf_map) and _verify.both_accept(pf_cof)
    assert _verify.both_accept(pf_pair) and (pf_pair[-288:-144] == final_rec[0]).all()
    nio = int(pf_map[3])
    assert (pf_map[-nio * 48:].reshape(nio, 48)[: n - 1] == map_recs).all()
    # the cleared points the SIPP statement's B consists of are the outputs the cofactor proof binds
    nio = int(pf_cof[3])
    outs = pf_cof[-nio * 104:].reshape(nio, 104)[n - 1: 2 * (n - 1), 72:]
    assert (outs == B[: n - 1]).all()
    inst = sipp_amd.Instance([x.shape[0] for x in ios])
    try:
        proofs = inst.prove(ios)
    finally:
        inst.close()
    for kind in range(3):
        assert _verify.both_accept(proofs[kind])
        nio = int(proofs[kind][3])
        w = ios[kind].shape[1]
        assert (proofs[kind][-nio * w:].reshape(nio, w)[: ios[kind].shape[0]] == ios[kind]).all()
