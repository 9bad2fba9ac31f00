"""Both verifiers over one proof: the library's own (sipp_stark_verify: sipp_amd/csrc/verify.cpp, what a user of the library runs --
starky's verify_stark_proof / data.verify of reference src/verifier_circuit.rs:254) and the CPU oracle's (oracle/stark.c, the checker).
A proof counts as accepted only if BOTH accept; a refusal must come from both, at the same stage."""
from tests import _oracle


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
