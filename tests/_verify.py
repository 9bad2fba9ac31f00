"""Both verifiers over one proof: the library's own (sipp_stark_verify: sipp_amd/csrc/verify.cpp, what a user of the library runs --
starky's verify_stark_proof / data.verify of reference src/verifier_circuit.rs:254) and the CPU oracle's (oracle/stark.c, the checker).
A proof counts as accepted only if BOTH accept; a refusal must come from both, at the same stage."""
import ctypes as C

import numpy as np

from tests import _oracle

u64p = C.POINTER(C.c_uint64)


def both_accept(proof, ocfg=None, cfg=None):
    import sipp_amd
    if cfg is None and ocfg is not None:
        cfg = sipp_amd.default_config()
        for name, _ in _oracle.OrcConfig._fields_:
            setattr(cfg, name, getattr(ocfg, name))
    a, b = sipp_amd.stark_verify(proof, cfg), _oracle.stark_verify(proof, ocfg)
    assert (a == 0) == (b == 0), "the verifiers disagree: library stage %d, oracle %d" % (a, b)
    return a == 0


def both_refuse(proof, ocfg=None, cfg=None):
    import sipp_amd
    if cfg is None and ocfg is not None:
        cfg = sipp_amd.default_config()
        for name, _ in _oracle.OrcConfig._fields_:
            setattr(cfg, name, getattr(ocfg, name))
    a, b = sipp_amd.stark_verify(proof, cfg), _oracle.stark_verify(proof, ocfg)
    assert a == -b, "the verifiers disagree: library stage %d, oracle %d" % (a, b)
    return a != 0


# ---- the generic proofs: oracle-side parameter structs in, the library's verdict (refusing stage, 0 = accepted) out ----
def lib_fri_params(ofp):
    import sipp_amd
    return sipp_amd.FriParams.from_buffer_copy(bytes(ofp))


def lib_plonk_verify(pf, cs_cap, p, ofp, circ, digest):
    import sipp_amd
    L = sipp_amd.lib()
    pf = np.ascontiguousarray(pf, dtype=np.uint64)
    cap = np.ascontiguousarray(cs_cap, dtype=np.uint64).reshape(-1)
    pp = sipp_amd.PlonkParams(p.num_routed_wires, p.max_degree, p.num_challenges)
    fp = lib_fri_params(ofp)
    cc = sipp_amd.PlonkCircuit.from_dict(circ)
    dg = np.asarray(digest, dtype=np.uint64)
    reason = C.c_int(0)
    rc = L.sipp_plonk_verify_gates(pf.ctypes.data_as(u64p), len(pf), cap.ctypes.data_as(u64p), C.byref(pp), C.byref(fp), C.byref(cc),
                                   dg.ctypes.data_as(u64p), C.byref(reason))
    assert rc in (0, -9), rc
    return reason.value


def lib_fri_verify(proof, caps, ncols, n_salt, batches, log_n, ofp, och):
    import sipp_amd
    L = sipp_amd.lib()
    proof = np.ascontiguousarray(proof, dtype=np.uint64)
    caps = [np.ascontiguousarray(c, dtype=np.uint64).reshape(-1) for c in caps]
    cp = (u64p * len(caps))(*[c.ctypes.data_as(u64p) for c in caps])
    nc = (C.c_uint32 * len(caps))(*ncols)
    ns = (C.c_uint32 * len(caps))(*n_salt)
    arr = (sipp_amd._lib.FriBatch * len(batches))()
    keep = []
    for i, (pt, ranges) in enumerate(batches):
        r = (sipp_amd._lib.PolyRange * len(ranges))(*[sipp_amd._lib.PolyRange(*x) for x in ranges])
        keep.append(r)
        arr[i].point[0], arr[i].point[1] = int(pt[0]), int(pt[1])
        arr[i].n_ranges = len(ranges)
        arr[i].ranges = r
    ch = sipp_amd.Challenger.from_buffer_copy(bytes(och))
    fp = lib_fri_params(ofp)
    reason = C.c_int(0)
    rc = L.sipp_fri_verify_openings(proof.ctypes.data_as(u64p), len(proof), cp, nc, ns, len(caps), arr, len(batches), log_n, C.byref(fp),
                                    C.byref(ch), C.byref(reason))
    assert rc in (0, -9), rc
    return reason.value, bytes(ch)
