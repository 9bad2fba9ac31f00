"""Generic PolynomialBatch commitments and FRI opening proofs through the C ABI (sipp_commit_batch_ex,
sipp_fri_prove_openings) against oracle/fri.c: caps, the complete flat proof and the challenger state after the proof are
identical word for word, for blowup 2 / 4 / 8, arity 2 .. 16 (constant and mixed), salted oracles and both PoW rules."""
import ctypes as C

import numpy as np
import pytest

from tests import _oracle, _verify
from tests.test_oracle_fri_generic import random_instance

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import sipp_amd
    c = sipp_amd.Ctx(workspace_bytes=2 << 30)
    yield c
    c.close()


def to_params(fp):
    import sipp_amd
    p = sipp_amd.FriParams()
    for f in ("rate_bits", "cap_height", "pow_bits", "num_queries", "pow_rule", "hiding", "n_rounds"):
        setattr(p, f, getattr(fp, f))
    for i in range(fp.n_rounds):
        p.arity_bits[i] = fp.arity_bits[i]
    return p


def gpu_challenger(seed):
    """the same transcript prefix as _oracle.challenger(seed): observed on the CPU oracle, copied as plain data"""
    import sipp_amd
    o = _oracle.challenger(seed)
    g = sipp_amd.Challenger()
    for i in range(12):
        g.state[i] = o.state[i]
    for i in range(8):
        g.in_buf[i], g.out_buf[i] = o.in_buf[i], o.out_buf[i]
    g.n_in, g.n_out = o.n_in, o.n_out
    return g, o


CONFIGS = [
    # log_n, rate_bits, cap_height, arity (bits or list), final_poly_bits, pow_rule, queries
    (10, 1, 4, 4, 5, 0, 9),
    (10, 2, 3, 3, 4, 0, 7),
    (11, 3, 4, 4, 5, 1, 6),       # plonky2's standard_recursion_config blowup and arity
    (10, 1, 0, 1, 5, 0, 5),       # arity 2: unhashed 4-word FRI leaves
    (12, 3, 2, [3, 1, 2, 4], 0, 0, 5),
    (13, 2, 4, 4, 5, 0, 8),
]


@pytest.mark.parametrize("log_n,rate_bits,cap_height,arity,final_poly_bits,pow_rule,nq", CONFIGS)
def test_generic_opening_proof_identical_to_oracle(ctx, log_n, rate_bits, cap_height, arity, final_poly_bits, pow_rule, nq):
    from sipp_amd._lib import to_device
    oracles, batches = random_instance(1000 + log_n + rate_bits, log_n, rate_bits, cap_height, ncols=(9, 6), salted=(False, True))
    if isinstance(arity, list):
        fp = _oracle.fri_params(rate_bits=rate_bits, cap_height=cap_height, pow_bits=9, num_queries=nq, pow_rule=pow_rule, hiding=1, arities=arity)
    else:
        fp = _oracle.fri_params(rate_bits=rate_bits, cap_height=cap_height, pow_bits=9, num_queries=nq, pow_rule=pow_rule, hiding=1,
                                arity_bits=arity, final_poly_bits=final_poly_bits, degree_bits=log_n)
    gch, och = gpu_challenger([7, 7, 7])
    ref = _oracle.fri_prove_openings(oracles, batches, log_n, fp, och)
    # commit the same data on the device: oracle 0 from values (re-derived from its coefficients), oracle 1 from coefficients
    devs, keep = [], []
    for k, o in enumerate(oracles):
        coeffs = o.coeffs
        if k == 0:
            vals = coeffs.copy()
            L = _oracle.load()
            for c in range(vals.shape[0]):
                L.orc_fft(vals[c], log_n)
            data, from_coeffs = vals, False
        else:
            data, from_coeffs = coeffs, True
        salt = None if o.salt is None else to_device(o.salt)
        od, cap, bufs = ctx.commit_ex(to_device(data), log_n, rate_bits, cap_height, from_coeffs=from_coeffs, salt=salt)
        assert (cap == o.cap).all(), k
        devs.append(od)
        keep.append(bufs)
    got = ctx.fri_prove_openings(devs, batches, log_n, to_params(fp), gch)
    assert len(got) == len(ref), (len(got), len(ref))
    diff = np.nonzero(got != ref)[0]
    assert diff.size == 0, "first mismatch at word %d of %d" % (diff[0], len(ref))
    # the transcript continues identically on both sides
    assert [gch.state[i] for i in range(12)] == [och.state[i] for i in range(12)]
    assert (gch.n_in, gch.n_out) == (och.n_in, och.n_out)
    assert _oracle.fri_verify_openings(got, [o.cap for o in oracles], [o.ncols for o in oracles], [o.n_salt for o in oracles], batches,
                                       log_n, fp, _oracle.challenger([7, 7, 7])) == 0
    stage, _ = _verify.lib_fri_verify(got, [o.cap for o in oracles], [o.ncols for o in oracles], [o.n_salt for o in oracles], batches, log_n, fp,
                                      _oracle.challenger([7, 7, 7]))
    assert stage == 0                        # the library's own verifier (sipp_fri_verify_openings)


def test_generic_api_argument_errors(ctx):
    import sipp_amd
    from sipp_amd._lib import to_device
    oracles, batches = random_instance(5, 10, 1, 2, ncols=(3, 2), salted=(False, False))
    data = to_device(oracles[0].coeffs)
    od, cap, bufs = ctx.commit_ex(data, 10, 1, 2, from_coeffs=True)
    fp = _oracle.fri_params(rate_bits=1, cap_height=2, pow_bits=4, num_queries=3, arity_bits=4, final_poly_bits=5, degree_bits=10)
    g, _ = gpu_challenger([])
    with pytest.raises(sipp_amd.SippError) as e:       # a range outside its oracle
        ctx.fri_prove_openings([od], [((3, 5), [(0, 0, 4)])], 10, to_params(fp), g)
    assert e.value.code == -1
    bad = to_params(fp)
    bad.arity_bits[0] = 5
    with pytest.raises(sipp_amd.SippError) as e:
        ctx.fri_prove_openings([od], [((3, 5), [(0, 0, 3)])], 10, bad, g)
    assert e.value.code == -7
    pf = ctx.fri_prove_openings([od], [((3, 5), [(0, 0, 3)])], 10, to_params(fp), g)     # the ctx is still usable
    assert pf[0] == 0x5349505046524931
