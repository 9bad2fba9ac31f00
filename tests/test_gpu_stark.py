"""GPU STARK prover through the C ABI vs the CPU oracle: the flat proofs must be identical word for word
(trace cap, Z cap, quotient cap, openings, FRI caps, final poly, PoW witness, every query opening), and the
oracle's verifier must accept them."""
import numpy as np
import pytest

from tests import _oracle, _verify

pytestmark = pytest.mark.gpu

SECTIONS = ["header", "trace_cap", "z_cap", "quotient_cap", "openings", "fri", "queries/public inputs"]


def locate(pf, pos):
    W, P = int(pf[4]), int(pf[5])
    bounds = [16, 16 + 64, 16 + 128, 16 + 192, 16 + 192 + 2 * (2 * W + 2 * P + 4)]
    rounds, flen = int(pf[8]), int(pf[9])
    bounds.append(bounds[-1] + rounds * 64 + 2 * flen + 1)
    for name, b in zip(SECTIONS, bounds):
        if pos < b:
            return name
    return SECTIONS[-1]


@pytest.fixture(scope="module")
def ctx():
    import sipp_amd
    c = sipp_amd.Ctx(workspace_bytes=8 << 30)
    yield c
    c.close()


@pytest.fixture(scope="module")
def ios4():
    d = np.load("tests/golden/sipp_n4_ios.npz")
    return d["g1"], d["g2"], d["fq12"]


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_proof_identical_to_oracle_and_verifies(ctx, ios4, kind):
    ios = ios4[kind]
    ref = _oracle.stark_prove(kind, ios)
    got = ctx.prove(kind, ios)
    assert len(got) == len(ref)
    diff = np.nonzero(got != ref)[0]
    assert diff.size == 0, "first mismatch at word %d (%s)" % (diff[0], locate(ref, int(diff[0])))
    assert _verify.both_accept(got)


def test_full_size_n128_proofs_verify(ctx):
    """BASELINE config n = 128 (127 / 127 / 14 IO records; N = 2^16, 2^16, 2^13): too large for the CPU prover in a test,
    so parity is checked through size-independent properties: the oracle's verifier accepts each GPU proof (all
    constraints at zeta, FRI, 84 query openings against the caps) and the public inputs are exactly the IO records."""
    d = np.load("tests/golden/sipp_n128_ios.npz")
    for kind, key in ((0, "g1"), (1, "g2"), (2, "fq12")):
        ios = d[key]
        pf = ctx.prove(kind, ios)
        assert _verify.both_accept(pf), key
        nio = int(pf[3])
        pis = pf[-nio * ios.shape[1]:].reshape(nio, ios.shape[1])
        assert (pis[: ios.shape[0]] == ios).all()
        assert (pis[ios.shape[0]:] == ios[-1]).all()      # padding repeats the last record
        bad = pf.copy()
        bad[16 + 3] ^= 1                                   # one bit of the trace cap
        assert _verify.both_refuse(bad)


def test_full_size_n128_proofs_equal_the_oracle_digests(ctx):
    """word-for-word parity at the BASELINE size: the oracle's n = 128 proofs take minutes of CPU, so their sha256
    digests are committed (tools/gen_golden.py digests128) and the GPU proofs must hash to the same values."""
    import hashlib
    import json
    gold = json.load(open("tests/golden/proof_digests_n128.json"))
    d = np.load("tests/golden/sipp_n128_ios.npz")
    for kind, key, src in ((0, "g1", "g1"), (1, "g2", "g2"), (2, "fq12", "fq12"), (4, "g1_hardened", "g1"), (5, "g2_hardened", "g2")):
        pf = ctx.prove(kind, d[src])       # kinds 4 / 5: the hardened AIRs bench.py's headline proves, over the same IO records
        assert len(pf) == gold[key]["words"], key
        assert hashlib.sha256(pf.tobytes()).hexdigest() == gold[key]["sha256"], key


def test_fallback_kernel_routes_give_the_same_proofs(ios4):
    """the two non-default kernel routes (sipp_ctx_set_kernel_routes; until round 5 environment switches) stay word for
    word equal to the oracle: openings by one block per column (traces >= 1024 rows), and columns of 2^13 rows through the
    whole-column-in-LDS transform (the Fq12 STARK of n = 128: the committed digest)"""
    import hashlib
    import json
    import sipp_amd
    L = sipp_amd.lib()
    c = sipp_amd.Ctx(workspace_bytes=L.sipp_workspace_bytes(2, 16))
    try:
        assert L.sipp_ctx_set_kernel_routes(c.h, 8) == -1                 # unknown bit (4 = SIPP_ROUTE_WITNESS_NO_GRAPH: tests/test_gpu_plonk.py)
        assert L.sipp_ctx_set_kernel_routes(c.h, 3) == 0
        for kind in (0, 2):
            pf = c.prove(kind, ios4[kind])
            ref = _oracle.stark_prove(kind, ios4[kind])
            assert len(pf) == len(ref) and (pf == ref).all(), kind
        gold = json.load(open("tests/golden/proof_digests_n128.json"))["fq12"]
        pf = c.prove(2, np.load("tests/golden/sipp_n128_ios.npz")["fq12"])
        assert int(pf[2]) == 13 and hashlib.sha256(pf.tobytes()).hexdigest() == gold["sha256"]
    finally:
        c.close()


def test_misaligned_device_buffers_are_refused_by_the_tree_transforms():
    """the tree sweeps move 16 bytes per lane; an 8-byte-aligned view of a device buffer (a tensor slice at an odd u64
    offset) is SIPP_E_BADARG through the public building block sipp_lde_batch, not left to the unaligned-access mode"""
    import torch
    import sipp_amd
    L = sipp_amd.lib()
    c = sipp_amd.Ctx(workspace_bytes=1 << 30)
    try:
        n, ncols = 1 << 15, 4
        buf = torch.zeros(ncols * n + 2, dtype=torch.int64, device="cuda")
        co = torch.zeros(ncols * n + 2, dtype=torch.int64, device="cuda")
        lde = torch.zeros(2 * ncols * n + 2, dtype=torch.int64, device="cuda")
        assert L.sipp_lde_batch(c.h, buf.data_ptr(), co.data_ptr(), lde.data_ptr(), ncols, 15) == 0
        assert L.sipp_lde_batch(c.h, buf.data_ptr() + 8, co.data_ptr(), lde.data_ptr(), ncols, 15) == -1
        assert L.sipp_lde_batch(c.h, buf.data_ptr(), co.data_ptr(), lde.data_ptr() + 8, ncols, 15) == -1
        assert L.sipp_lde_batch(c.h, buf.data_ptr(), co.data_ptr(), lde.data_ptr(), ncols, 15) == 0       # the ctx survives
    finally:
        c.close()


def test_error_behaviour(ctx, ios4):
    """the C ABI's error contract (include/sipp_hip.h): wrong claimed output -> SIPP_E_WITNESS (the CPU restatement
    refuses the same record), short buffer -> SIPP_E_BUFSZ, bad arguments -> SIPP_E_BADARG; the ctx stays usable."""
    import ctypes as C
    import sipp_amd
    L = sipp_amd.lib()
    for kind in (0, 1, 2):
        ios = ios4[kind].copy()
        ios[1, -1] ^= 1
        with pytest.raises(sipp_amd.SippError) as e:
            ctx.prove(kind, ios)
        assert e.value.code == -8
        with pytest.raises(RuntimeError):
            _oracle.stark_prove(kind, ios)
    ios = np.ascontiguousarray(ios4[0])
    out = np.zeros(16, dtype=np.uint64)
    n = C.c_size_t()
    assert L.sipp_g1_exp_prove(ctx.h, ios.ctypes.data, ios.shape[0], out.ctypes.data, 16, C.byref(n)) == -4
    assert L.sipp_g1_exp_prove(ctx.h, None, 3, out.ctypes.data, 16, C.byref(n)) == -1
    assert L.sipp_g1_exp_prove(ctx.h, ios.ctypes.data, 0, out.ctypes.data, 16, C.byref(n)) == -1
    assert L.sipp_proof_size(ctx.h, 7, 3) == 0
    assert _verify.both_accept(ctx.prove(0, ios4[0]))     # still works afterwards


def test_n256_instance_of_the_reference_test_verifies():
    """the size the reference's own `test_sipp_circuit` runs today (src/verifier_circuit.rs:199-200: log_n = 8): 255 G1 / 255 G2 /
    16 Fq12 records, N = 2^17 / 2^17 / 2^13, the three proofs of ONE instance on three concurrent ctxs; every proof through the
    oracle's verifier, public inputs = the records"""
    import sipp_amd
    d = np.load("tests/golden/sipp_n256_ios.npz")
    ios = [d[k] for k in ("g1", "g2", "fq12")]
    assert [a.shape for a in ios] == [(255, 56), (255, 104), (16, 296)]
    inst = sipp_amd.Instance([a.shape[0] for a in ios])
    try:
        proofs = inst.prove(ios)
    finally:
        inst.close()
    for kind, (pf, rec) in enumerate(zip(proofs, ios)):
        assert int(pf[1]) == kind and int(pf[2]) == (17, 17, 13)[kind]
        assert _verify.both_accept(pf), kind
        nio = int(pf[3])
        assert (pf[-nio * rec.shape[1]:].reshape(nio, rec.shape[1])[: rec.shape[0]] == rec).all()


def _instance_proofs_verify(n, hardened, log_n, digests_required=False):
    """the three sub-proofs of the n-pairing fixture through sipp_amd.Instance exactly as bench.py's `io_sharded` leg builds it
    (hardened=True: API kinds 4 / 5, the headline variant; three ctxs, or ONE arena with the proofs back to back where three do
    not fit the card -- hardened n = 4096), every proof through the oracle's verifier, public inputs = the records"""
    import sipp_amd
    d = np.load("tests/golden/sipp_n%d_ios.npz" % n)
    ios = [d[k] for k in ("g1", "g2", "fq12")]
    inst = sipp_amd.Instance([a.shape[0] for a in ios], hardened=hardened)
    single = inst.single_ctx
    checked_digests = []
    try:
        proofs = [p.copy() for p in inst.prove(ios)]
    finally:
        inst.close()
    for k, (pf, rec) in enumerate(zip(proofs, ios)):
        assert int(pf[1]) == (k + 4 if hardened and k < 2 else k), k
        nio = int(pf[3])
        assert nio >= rec.shape[0] and (nio & (nio - 1)) == 0
        if k < 2:
            assert int(pf[2]) == log_n and nio == n
        assert _verify.both_accept(pf), (k, hardened)
        pis = pf[-nio * rec.shape[1]:].reshape(nio, rec.shape[1])
        assert (pis[: rec.shape[0]] == rec).all() and (pis[rec.shape[0]:] == rec[-1]).all()
        bad = pf.copy()
        bad[16 + 64 + 5] ^= 1                              # one bit of the Z cap
        assert _verify.both_refuse(bad)
        # WORD-FOR-WORD parity at the large configs: the sha256 of the oracle's proof of the same records, computed
        # offline (tools/gen_golden.py digests_large: the CPU prover takes up to an hour and 40 GB per proof)
        key = "n%d.%s" % (n, ("g1", "g2", "fq12")[k] + ("_hardened" if hardened and k < 2 else ""))
        gold = _large_digests().get(key)
        if gold is not None:
            import hashlib
            assert len(pf) == gold["words"] and hashlib.sha256(pf.tobytes()).hexdigest() == gold["sha256"], key
            checked_digests.append(key)
    if digests_required:
        assert len(checked_digests) == 3, checked_digests
    return single


def _large_digests():
    import json
    return json.load(open("tests/golden/proof_digests_large.json"))


@pytest.mark.parametrize("hardened", [False, True])
def test_large_n1024_proofs_verify(hardened):
    """deep traces (BASELINE config 3): n = 1024 -> 1023 G1 / 1023 G2 / 20 Fq12 IO records, N = 2^19 / 2^19 / 2^14 rows,
    2^20-leaf trees, tree-of-rings transforms of 2^19 / 2^20 points, the u16 lookup table.  Plain kinds: one ctx per kind; hardened
    kinds 4 / 5 (the variant bench.py's `value` and `io_sharded` legs prove): through sipp_amd.Instance.  Checked through the
    oracle's verifier and the public inputs."""
    import sipp_amd
    if hardened:
        # BASELINE configs[2]: a digest-equality assertion for all three proofs, not only a verifier pass
        assert _instance_proofs_verify(1024, True, 19, digests_required=True) is False       # three arenas fit the card
        return
    d = np.load("tests/golden/sipp_n1024_ios.npz")
    L = sipp_amd.lib()
    for kind, key in ((0, "g1"), (1, "g2"), (2, "fq12")):
        ios = d[key]
        c = sipp_amd.Ctx(workspace_bytes=L.sipp_workspace_bytes(kind, ios.shape[0]))
        try:
            pf = c.prove(kind, ios)
        finally:
            c.close()
        nio = int(pf[3])
        assert nio >= ios.shape[0] and (nio & (nio - 1)) == 0
        if kind < 2:
            assert int(pf[2]) == 19 and nio == 1024
        assert _verify.both_accept(pf), key
        assert (pf[-nio * ios.shape[1]:].reshape(nio, ios.shape[1])[: ios.shape[0]] == ios).all()


def test_workspace_too_small_fails_cleanly(ios4):
    """a ctx whose arena cannot hold the proof's buffers returns SIPP_E_NOMEM (no fault, no partial proof) and a
    properly sized ctx created afterwards still works"""
    import sipp_amd
    small = sipp_amd.Ctx(workspace_bytes=8 << 20)
    try:
        with pytest.raises(sipp_amd.SippError) as e:
            small.prove(1, ios4[1])
        assert e.value.code == -3      # SIPP_E_NOMEM
    finally:
        small.close()
    ok = sipp_amd.Ctx(workspace_bytes=sipp_amd.lib().sipp_workspace_bytes(1, ios4[1].shape[0]))
    try:
        assert _verify.both_accept(ok.prove(1, ios4[1]))
    finally:
        ok.close()


def test_edge_exponents_and_degenerate_inputs(ctx):
    """exp = 0 / 1 / r - 1 / high-bit patterns prove and match the oracle word for word; a degenerate record
    (x == offset, odd exponent: the first addition would be a doubling) fails loudly with SIPP_E_WITNESS."""
    import sipp_amd
    from tests.test_oracle_air import crafted_g1_records
    recs = crafted_g1_records()
    got = ctx.prove(0, recs)
    assert (got == _oracle.stark_prove(0, recs)).all()
    assert _verify.both_accept(got)
    from tests.test_oracle_air import crafted_g2_records, crafted_fq12_records
    for kind, r2 in ((1, crafted_g2_records()), (2, crafted_fq12_records())):
        got2 = ctx.prove(kind, r2)
        assert (got2 == _oracle.stark_prove(kind, r2)).all(), kind
    from oracle.py import bn254 as bn
    from oracle.py import sipp_native as sn
    x = bn.g1_mul(bn.G1, 77)
    bad = np.array([bn.g1_to_u32(x) + bn.g1_to_u32(x) + sn.exp_to_u32(3) + bn.g1_to_u32(bn.g1_mul(x, 4))], dtype=np.uint32)
    with pytest.raises(sipp_amd.SippError) as e:
        ctx.prove(0, bad)
    assert e.value.code == -8


@pytest.mark.parametrize("hardened", [False, True])
def test_max_size_n4096_proofs_verify(hardened):
    """largest BASELINE config: n = 4096 -> 4095 G1 / 4095 G2 / 24 Fq12 IO records, N = 2^21 rows, LDE 2^22 rows
    (five-sweep tree transforms), 82 GB / 160 GB arenas.  The oracle's verifier must accept every proof.  hardened: kinds 4 / 5
    through the single-arena Instance (179 GB, the three proofs back to back) exactly as bench.py's io_sharded["n=4096"] builds it."""
    import sipp_amd
    if hardened:
        assert _instance_proofs_verify(4096, True, 21) is True        # three arenas (276 GB) do not fit: one ctx
        return
    d = np.load("tests/golden/sipp_n4096_ios.npz")
    L = sipp_amd.lib()
    for kind, key in ((0, "g1"), (1, "g2"), (2, "fq12")):
        ios = d[key]
        c = sipp_amd.Ctx(workspace_bytes=L.sipp_workspace_bytes(kind, ios.shape[0]))
        try:
            pf = c.prove(kind, ios)
        finally:
            c.close()
        if kind < 2:
            assert int(pf[2]) == 21 and int(pf[3]) == 4096
        assert _verify.both_accept(pf), key


def test_concurrent_streams_are_deterministic():
    """bench.py drives one ctx (= HIP stream + arena) per sub-proof from its own host thread, and several instances in
    flight for the `pipelined` figure.  The proofs must not depend on that: two instances x three sub-proofs proved
    concurrently, twice, equal the proofs of a single ctx run serially, word for word."""
    from concurrent.futures import ThreadPoolExecutor
    import sipp_amd
    d = np.load("tests/golden/sipp_n8_ios.npz")
    ios = [d["g1"], d["g2"], d["fq12"]]
    L = sipp_amd.lib()
    serial_ctx = sipp_amd.Ctx(workspace_bytes=max(L.sipp_workspace_bytes(k, ios[k].shape[0]) for k in range(3)))
    want = [serial_ctx.prove(k, ios[k]) for k in range(3)]
    serial_ctx.close()
    groups = [[sipp_amd.Ctx(workspace_bytes=L.sipp_workspace_bytes(k, ios[k].shape[0])) for k in range(3)] for _ in range(2)]
    jobs = [(g, k) for g in range(2) for k in range(3)]
    with ThreadPoolExecutor(max_workers=6) as pool:
        for _ in range(2):
            got = list(pool.map(lambda gk: groups[gk[0]][gk[1]].prove(gk[1], ios[gk[1]]), jobs))
            for (g, k), pf in zip(jobs, got):
                assert len(pf) == len(want[k]) and (pf == want[k]).all(), (g, k)
    for row in groups:
        for c in row:
            c.close()
    assert _verify.both_accept(want[0])


def test_async_and_instance_calls_give_the_same_proofs(ctx, ios4):
    """sipp_prove_async + sipp_wait and sipp_instance_prove (three ctxs, three streams, the library's own worker threads)
    return exactly the proofs of the synchronous calls; a second job on a busy ctx and a wait without a job are refused."""
    import sipp_amd
    want = [ctx.prove(k, ios4[k]) for k in range(3)]
    got = ctx.prove_async(0, ios4[0]).wait()
    assert (got == want[0]).all()
    with pytest.raises(sipp_amd.SippError) as e:
        ctx.wait()
    assert e.value.code == -1
    ctx.prove_async(1, ios4[1])
    with pytest.raises(sipp_amd.SippError) as e:
        ctx.prove_async(0, ios4[0])
    assert e.value.code == -1
    assert (ctx.wait() == want[1]).all()
    inst = sipp_amd.Instance([a.shape[0] for a in ios4])
    try:
        for _ in range(2):
            proofs = inst.prove(ios4)
            for k in range(3):
                assert len(proofs[k]) == len(want[k]) and (proofs[k] == want[k]).all(), k
    finally:
        inst.close()


def test_async_error_is_reported_by_wait(ctx, ios4):
    """an unprovable IO record (claimed output off by one) surfaces as SIPP_E_WITNESS from sipp_wait, like the synchronous call"""
    import sipp_amd
    bad = ios4[0].copy()
    bad[0, -1] ^= 1
    ctx.prove_async(0, bad)
    with pytest.raises(sipp_amd.SippError) as e:
        ctx.wait()
    assert e.value.code == -8
    assert (ctx.prove(0, ios4[0])[:16] == _oracle.stark_prove(0, ios4[0])[:16]).all()   # the ctx is usable afterwards


@pytest.mark.parametrize("kind,num", [(0, 1), (0, 5), (1, 2), (1, 7), (2, 1), (2, 3)])
def test_ragged_io_counts_match_the_oracle(ctx, kind, num):
    """IO lists that are not a power of two (and a single record) are padded with copies of the last record on both sides:
    same proof word for word, and the public inputs show the padding."""
    d = np.load("tests/golden/sipp_n8_ios.npz")
    ios = d[("g1", "g2", "fq12")[kind]][:num]
    ref = _oracle.stark_prove(kind, ios)
    got = ctx.prove(kind, ios)
    assert len(got) == len(ref)
    diff = np.nonzero(got != ref)[0]
    assert diff.size == 0, "first mismatch at word %d (%s)" % (diff[0], locate(ref, int(diff[0])))
    nio = int(got[3])
    assert nio >= max(2, num) and nio & (nio - 1) == 0
    pis = got[-nio * ios.shape[1]:].reshape(nio, ios.shape[1])
    assert (pis[:num] == ios).all() and (pis[num:] == ios[-1]).all()


def test_config0_n8_instance_matches_the_oracle():
    """BASELINE configs[0] (n = 8, the reference's CPU-runnable plumbing case: 7 / 7 / 6 obligations): the whole instance through
    sipp_instance_prove equals the CPU oracle's three proofs word for word and verifies."""
    import sipp_amd
    d = np.load("tests/golden/sipp_n8_ios.npz")
    ios = [d["g1"], d["g2"], d["fq12"]]
    assert [a.shape[0] for a in ios] == [7, 7, 6]
    inst = sipp_amd.Instance([a.shape[0] for a in ios])
    try:
        proofs = [p.copy() for p in inst.prove(ios)]
    finally:
        inst.close()
    for kind in range(3):
        ref = _oracle.stark_prove(kind, ios[kind])
        assert len(proofs[kind]) == len(ref), kind
        diff = np.nonzero(proofs[kind] != ref)[0]
        assert diff.size == 0, "kind %d: first mismatch at word %d (%s)" % (kind, diff[0], locate(ref, int(diff[0])))
        assert _verify.both_accept(proofs[kind])


@pytest.mark.timeout(300)
def test_instance_with_an_unprovable_g2_record_fails_without_hanging(ios4):
    """G1 and Fq12 wait for G2's trace fill (the start gate of sipp_instance_prove): a G2 proof that fails before that
    point must still let them go.  The call reports SIPP_E_WITNESS and the instance stays usable."""
    import sipp_amd
    inst = sipp_amd.Instance([a.shape[0] for a in ios4])
    try:
        bad = [a.copy() for a in ios4]
        bad[1][0, -1] ^= 1
        with pytest.raises(sipp_amd.SippError) as e:
            inst.prove(bad)
        assert e.value.code == -8
        good = inst.prove(ios4)
        assert _verify.both_accept(good[1])
    finally:
        inst.close()


@pytest.mark.parametrize("seed", [11, 12])
def test_random_obligations_match_the_oracle(ctx, seed):
    """fresh obligations per seed (random points, random Fq12 elements, random exponents; nothing from the fixtures): outputs
    from sipp_exp_outputs, then every proof word for word against the oracle (which recomputes and checks the outputs itself)."""
    from oracle.py import bn254 as bn
    rng = np.random.default_rng(seed)

    def scalar():
        return int.from_bytes(rng.bytes(32), "little") % bn.R or 1

    def fq():
        return int.from_bytes(rng.bytes(32), "little") % bn.P

    g1 = [bn.g1_to_u32(bn.g1_mul(bn.G1, scalar())) + bn.g1_to_u32(bn.g1_mul(bn.G1, scalar())) +
          [(e >> (32 * i)) & 0xFFFFFFFF for e in [scalar()] for i in range(8)] + [0] * 16 for _ in range(3)]
    g2 = [bn.g2_to_u32(bn.g2_mul(bn.G2, scalar())) + bn.g2_to_u32(bn.g2_mul(bn.G2, scalar())) +
          [(e >> (32 * i)) & 0xFFFFFFFF for e in [scalar()] for i in range(8)] + [0] * 32 for _ in range(2)]
    f12 = [bn.f12_to_u32([fq() for _ in range(12)]) + bn.f12_to_u32([fq() for _ in range(12)]) +
           [(e >> (32 * i)) & 0xFFFFFFFF for e in [scalar()] for i in range(8)] + [0] * 96 for _ in range(2)]
    for kind, recs in ((0, g1), (1, g2), (2, f12)):
        ios = ctx.exp_outputs(kind, np.array(recs, dtype=np.uint32))
        ref = _oracle.stark_prove(kind, ios)          # raises if an output were wrong
        got = ctx.prove(kind, ios)
        assert len(got) == len(ref)
        diff = np.nonzero(got != ref)[0]
        assert diff.size == 0, "kind %d: first mismatch at word %d (%s)" % (kind, diff[0], locate(ref, int(diff[0])))


@pytest.mark.parametrize("cap_height,pow_bits,final_poly_bits,num_queries", [(2, 10, 4, 20), (5, 18, 6, 30), (0, 4, 5, 3)])
def test_non_default_stark_config_matches_the_oracle(ios4, cap_height, pow_bits, final_poly_bits, num_queries):
    """sipp_stark_config other than standard_fast_config (cap height, proof-of-work bits, final polynomial size, query
    count): the proof shape changes with it, and stays word for word the oracle's for the same config."""
    import sipp_amd
    cfg = sipp_amd.default_config()
    ocfg = _oracle.default_config()
    for c in (cfg, ocfg):
        c.cap_height, c.pow_bits, c.final_poly_bits, c.num_queries = cap_height, pow_bits, final_poly_bits, num_queries
    ctx = sipp_amd.Ctx(cfg=cfg, workspace_bytes=4 << 30)
    try:
        for kind in (0, 2):
            ref = _oracle.stark_prove(kind, ios4[kind], ocfg)
            got = ctx.prove(kind, ios4[kind])
            assert len(got) == len(ref), (kind, len(got), len(ref))
            diff = np.nonzero(got != ref)[0]
            assert diff.size == 0, "kind %d: first mismatch at word %d" % (kind, diff[0])
            assert _verify.both_accept(got, ocfg)
            assert int(got[7]) == cap_height and int(got[10]) == num_queries
    finally:
        ctx.close()


@pytest.mark.parametrize("rate_bits,arity_bits,final_poly_bits,pow_rule", [(2, 4, 5, 0), (3, 3, 4, 0), (1, 2, 3, 1), (2, 1, 6, 0), (3, 4, 5, 1)])
def test_other_blowups_arities_and_pow_rules_match_the_oracle(ios4, rate_bits, arity_bits, final_poly_bits, pow_rule):
    """blowup 4 / 8 (plonky2's outer circuit uses 8), reduction arity 2 / 4 / 8 and the second proof-of-work rule: the proof
    shape changes (LDE size, FRI layers, leaf widths) and stays word for word the oracle's"""
    import ctypes as C
    import sipp_amd
    cfg = sipp_amd.default_config()
    ocfg = _oracle.default_config()
    for c in (cfg, ocfg):
        c.rate_bits, c.arity_bits, c.final_poly_bits, c.pow_rule, c.num_queries, c.pow_bits = rate_bits, arity_bits, final_poly_bits, pow_rule, 11, 9
    L = sipp_amd.lib()
    for kind in (0, 2):
        ws = L.sipp_workspace_bytes_cfg(kind, ios4[kind].shape[0], C.byref(cfg))
        assert ws > L.sipp_workspace_bytes(kind, ios4[kind].shape[0]) or rate_bits == 1
        ctx = sipp_amd.Ctx(cfg=cfg, workspace_bytes=ws)
        try:
            ref = _oracle.stark_prove(kind, ios4[kind], ocfg)
            got = ctx.prove(kind, ios4[kind])
            assert len(got) == len(ref), (kind, len(got), len(ref))
            diff = np.nonzero(got != ref)[0]
            assert diff.size == 0, "kind %d: first mismatch at word %d of %d" % (kind, diff[0], len(ref))
            assert _verify.both_accept(got, ocfg)
        finally:
            ctx.close()


@pytest.mark.parametrize("fs_rule,lookup_rule", [(1, 0), (0, 1), (1, 1)])
def test_protocol_rules_match_the_oracle_word_for_word(ios4, fs_rule, lookup_rule):
    """sipp_stark_config.fs_rule / lookup_rule (include/sipp_hip.h): starky's recalled transcript order (challenger
    starts at the trace cap) and single-column permutation pairs under one challenge, as DATA next to pow_rule.  Full-size
    configuration (84 queries, 16 grinding bits), every plain kind and a hardened one: the GPU proof is the oracle's word for word
    and both verifiers accept it under the same rules.  Rule 0 / 0 stays the default (the committed digests do not move)."""
    import ctypes as C
    import sipp_amd
    from oracle.py import stark_verify as sv
    cfg = sipp_amd.default_config()
    ocfg = _oracle.default_config()
    for c in (cfg, ocfg):
        c.fs_rule, c.lookup_rule = fs_rule, lookup_rule
    L = sipp_amd.lib()
    # every kind under BOTH rules together (the upstream-shaped mode); each single rule on a curve kind and the Fq12 kind (round 6: the
    # suite's time budget -- the rules act in the shared prover, not per AIR)
    for kind in ((0, 1, 2, 4) if (fs_rule and lookup_rule) else (0, 2)):
        ios = ios4[kind & 3]
        ctx = sipp_amd.Ctx(cfg=cfg, workspace_bytes=L.sipp_workspace_bytes_cfg(kind, ios.shape[0], C.byref(cfg)))
        try:
            if kind >= 4:
                ctx._ck(L.sipp_ctx_set_hardened(ctx.h, 1), "set_hardened")
            got = ctx.prove(kind & 3, ios)
        finally:
            ctx.close()
        ref = _oracle.stark_prove(kind, ios, ocfg)
        assert int(got[1]) == kind and int(got[15]) == (fs_rule | lookup_rule << 1)
        assert len(got) == len(ref)
        diff = np.nonzero(got != ref)[0]
        assert diff.size == 0, "kind %d: first mismatch at word %d (%s)" % (kind, diff[0], locate(ref, int(diff[0])))
        assert _verify.both_accept(got, ocfg)
        assert _oracle.stark_verify(got) == -102 and sipp_amd.stark_verify(got) == 102     # not a proof under the default rules
        if kind == 0:
            assert sv.verify(got, dict(fs_rule=fs_rule, lookup_rule=lookup_rule)) is None
    bad = sipp_amd.default_config()
    bad.fs_rule = 2
    h = C.c_void_p()
    assert L.sipp_ctx_create(C.byref(h), 0, C.byref(bad), 1 << 20) == -7       # SIPP_E_UNSUPPORTED


def test_g2_cofactor_clearing_runs_through_the_g2_exp_stark(ctx):
    """SURVEY 8f rank 4 (reference src/bin/bls_aggregation.rs:65,103-106), the arithmetic half: messages mapped to E'(Fp2) are
    multiplied by the cofactor 2p - r.  That is a G2 obligation out = offset + [exp] x with exp = cofactor (254 bits) on points
    OUTSIDE the r-torsion -- the same records, AIR and prover as the SIPP G2 list.  Outputs from sipp_exp_outputs are checked
    against the Python restatement (and land in the r-torsion once the offset is removed); the proof equals the oracle's word
    for word.  The map to the curve itself is not in the reference tree and is not restated."""
    from oracle.py import bn254 as bn
    h = bn.G2_COFACTOR
    offset = bn.g2_mul(bn.G2, 0x5151)
    pts = [bn.g2_twist_point(seed) for seed in (11, 12, 13)]
    recs = [bn.g2_to_u32(p) + bn.g2_to_u32(offset) + [(h >> (32 * i)) & 0xFFFFFFFF for i in range(8)] + [0] * 32 for p in pts]
    ios = ctx.exp_outputs(1, np.array(recs, dtype=np.uint32))
    for p, rec in zip(pts, ios):
        w = [int(x) for x in rec[72:104]]
        out = ((bn.u32_to_fq(w[0:8]), bn.u32_to_fq(w[8:16])), (bn.u32_to_fq(w[16:24]), bn.u32_to_fq(w[24:32])))
        cleared = bn.g2_add(out, bn.g2_neg(offset))
        assert cleared == bn.g2_clear_cofactor(p)
        assert bn.g2_mul(p, bn.R) is not None and bn.g2_mul(cleared, bn.R) is None
    ref = _oracle.stark_prove(1, ios)
    got = ctx.prove(1, ios)
    assert len(got) == len(ref) and (got == ref).all()
    assert _verify.both_accept(got)


def test_instance_queue_gives_the_single_instance_proofs(ios4):
    """sipp_instances_prove: five instances (alternating the n = 4 and a permuted copy of its obligation lists) through two slots
    of three ctxs -- every proof equals what one Instance produces for the same lists; a bad record fails only its instance."""
    import sipp_amd
    other = [np.ascontiguousarray(a[::-1]) for a in ios4]
    single = sipp_amd.Instance([a.shape[0] for a in ios4])
    try:
        want = {0: [p.copy() for p in single.prove(ios4)], 1: [p.copy() for p in single.prove(other)]}
    finally:
        single.close()
    q = sipp_amd.InstanceQueue([a.shape[0] for a in ios4], in_flight=2)
    try:
        got = q.prove([ios4, other, ios4, other, ios4])
        assert len(got) == 5
        for i, proofs in enumerate(got):
            for k in range(3):
                assert len(proofs[k]) == len(want[i % 2][k]) and (proofs[k] == want[i % 2][k]).all(), (i, k)
        bad = [a.copy() for a in ios4]
        bad[0][0, -1] ^= 1                       # a wrong claimed output in the G1 list of the second instance
        with pytest.raises(sipp_amd.SippError):
            q.prove([ios4, bad, ios4])
        again = q.prove([ios4])                   # the slots are usable afterwards
        assert all((again[0][k] == want[0][k]).all() for k in range(3))
    finally:
        q.close()


@pytest.mark.parametrize("kind", [0, 1])
def test_offset_a_small_multiple_of_the_base_point(ctx, kind):
    """Found by scripts/stress_parity.py (round 2): with offset = 3 x and exponent bits 0 and 1 set, the parallel scan of the
    trace generator adds two EQUAL partial sums (x + 2 x and the offset) although the AIR's sequential chain 3x -> 4x -> 6x
    never meets R = +-P.  The Jacobian addition of the scan is complete now; what is refused is exactly what the CPU
    restatement refuses (rows without a witness), and everything else is proved word for word."""
    import sipp_amd
    from oracle.py import bn254 as bn
    mul, G, to_u32 = (bn.g1_mul, bn.G1, bn.g1_to_u32) if kind == 0 else (bn.g2_mul, bn.G2, bn.g2_to_u32)
    neg = bn.g1_neg if kind == 0 else bn.g2_neg
    pad = 16 if kind == 0 else 32

    def rec(xk, offk, e):
        off = mul(G, abs(offk))
        if offk < 0:
            off = neg(off)
        return to_u32(mul(G, xk)) + to_u32(off) + [(e >> (32 * i)) & 0xFFFFFFFF for i in range(8)] + [0] * pad

    allones = (1 << 256) - 1
    provable = [rec(1, 3, allones - 0x330e4), rec(1, 3, 0b1011), rec(1, 5, 0b10111), rec(2, 6, 0x3), rec(1, 7, allones)]
    for r in provable:                      # one record per proof: a refusal cannot hide behind another record
        ios = ctx.exp_outputs(kind, np.array([r], dtype=np.uint32))
        ref = _oracle.stark_prove(kind, ios)
        got = ctx.prove(kind, ios)
        assert len(got) == len(ref) and (got == ref).all()
    # rows without a witness: both sides refuse (2x + [2] x doubles on an add row; -3x + x + 2x passes through infinity)
    for r in (rec(1, 2, 0b10), rec(1, -3, 0b11)):
        arr = np.array([r], dtype=np.uint32)
        with pytest.raises(RuntimeError):
            _oracle.stark_prove(kind, arr)
        with pytest.raises(sipp_amd.SippError):
            ctx.prove(kind, ctx.exp_outputs(kind, arr))


def test_config_that_folds_below_sixteen_values_is_declined_before_any_work(ios4):
    """Found by scripts/stress_stark_cfg.py (round 2): arity 2 with final_poly_bits 0 and a low cap folds the last committed FRI layers below
    the 16 values the layer kernels work on; the prove call used to fail in a transform wrapper in the middle of the proof
    (SIPP_E_BADARG "ntt: log_n out of range").  It is declined up front with SIPP_E_UNSUPPORTED now, and the ctx stays usable."""
    import sipp_amd
    cfg = sipp_amd.default_config()
    cfg.arity_bits, cfg.final_poly_bits, cfg.rate_bits, cfg.cap_height = 1, 0, 2, 0
    ctx = sipp_amd.Ctx(cfg=cfg, workspace_bytes=4 << 30)
    try:
        with pytest.raises(sipp_amd.SippError) as e:
            ctx.prove(0, ios4[0])
        assert e.value.code == -7 and "16 values" in str(e.value)
    finally:
        ctx.close()
    ocfg = _oracle.default_config()
    ocfg.arity_bits, ocfg.final_poly_bits, ocfg.rate_bits, ocfg.cap_height = 1, 0, 2, 0
    assert _verify.both_accept(_oracle.stark_prove(0, ios4[0], ocfg), ocfg)     # the protocol itself allows it


@pytest.mark.parametrize("kind", [0, 1])
def test_points_off_the_curve_are_refused_by_both_sides(ctx, kind):
    """The chord / tangent rules are a group law only on one curve; the GPU generator sums the selected powers in another order
    than the AIR's sequential chain, so for a base point and an offset on DIFFERENT curves y^2 = x^3 + b the two would disagree.
    Both sides refuse any record whose x or offset is not on E(Fp) / E'(Fp2) (SIPP_E_WITNESS / a failed trace build)."""
    import sipp_amd
    from oracle.py import bn254 as bn
    if kind == 0:
        good, off, pad = bn.g1_to_u32(bn.g1_mul(bn.G1, 77)), bn.fq_to_u32(5) + bn.fq_to_u32(7), 16
    else:
        good, pad = bn.g2_to_u32(bn.g2_mul(bn.G2, 77)), 32
        off = bn.fq_to_u32(5) + bn.fq_to_u32(1) + bn.fq_to_u32(7) + bn.fq_to_u32(2)
    e = [(0b1011 >> (32 * i)) & 0xFFFFFFFF for i in range(8)]
    for rec in (off + good + e + [0] * pad, good + off + e + [0] * pad):
        arr = np.array([rec], dtype=np.uint32)
        with pytest.raises(RuntimeError):
            _oracle.stark_prove(kind, arr)
        with pytest.raises(sipp_amd.SippError) as err:
            ctx.exp_outputs(kind, arr)
        assert err.value.code == -8
        with pytest.raises(sipp_amd.SippError) as err:
            ctx.prove(kind, arr)
        assert err.value.code == -8
