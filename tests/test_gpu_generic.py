"""GPU parity tests of the generic layer, through the C ABI, against the CPU oracle (bit-exact)."""
import numpy as np
import pytest

from tests import _oracle
from tests._oracle import P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import sipp_amd
    c = sipp_amd.Ctx(workspace_bytes=2 << 30)
    yield c
    c.close()


def dev(a):
    from sipp_amd._lib import to_device
    return to_device(a)


def host(t):
    from sipp_amd._lib import to_host
    return to_host(t)


def test_poseidon_kats_and_random(ctx):
    from tests.test_oracle_generic import KAT
    rng = np.random.default_rng(1)
    states = np.concatenate([np.array([k[0] for k in KAT], dtype=np.uint64), _oracle.rand_field(rng, (1000, 12))])
    got = host(ctx.poseidon_permute(dev(states)))
    for i, (_, out) in enumerate(KAT):
        assert [int(x) for x in got[i]] == out
    for i in range(3, states.shape[0]):
        assert (got[i] == _oracle.permute(states[i])).all(), i


@pytest.mark.parametrize("log_n", [4, 5, 8, 12, 13, 14, 17, 20, 22])
def test_ntt_matches_oracle(ctx, oracle, log_n):
    rng = np.random.default_rng(log_n)
    ncols = 3 if log_n < 18 else 2 if log_n < 21 else 1     # 2^22: three-pass plan (LDE size of the n = 4096 config)
    a = _oracle.rand_field(rng, (ncols, 1 << log_n))
    a[0, :4] = [0, 1, P - 1, P - 2]
    d = dev(a)
    got = host(ctx.ntt(d, log_n))
    for c in range(ncols):
        ref = a[c].copy()
        oracle.orc_fft(ref, log_n)
        assert (got[c] == ref).all(), (log_n, c)
    back = host(ctx.ntt(d, log_n, inverse=True))
    assert (back == a).all()


@pytest.mark.parametrize("log_n,ncols", [(4, 1), (6, 11), (9, 3), (10, 4), (11, 3), (12, 9), (13, 17), (14, 5), (15, 3), (16, 5),
                                         (17, 2), (18, 2), (19, 2), (20, 1), (21, 1), (22, 1)])
def test_commit_matches_oracle(ctx, log_n, ncols):
    """PolynomialBatch::from_values on the device -- whole-column fused kernel (2^10..2^14), the tree-of-rings sweeps (2^15 and above), pass-by-pass path below 2^10 --
    and the per-subtree Merkle kernels: coefficients, every LDE cell, EVERY tree level and the cap equal the oracle's"""
    rng = np.random.default_rng(100 + log_n)
    vals = _oracle.rand_field(rng, (ncols, 1 << log_n))
    ref = _oracle.Batch(vals, log_n)
    coeffs, lde, tree, cap = ctx.commit(dev(vals), log_n)
    assert (host(coeffs) == ref.coeffs).all()
    assert (host(lde).T == ref.leaves).all()          # column-major leaf order == transposed leaves
    m = 2 << log_n
    t = host(tree)
    off = 0
    for lvl in range(log_n + 1 - 4 + 1):               # leaves up to the cap level (cap height 4)
        cnt = m >> lvl
        assert (t[off:off + cnt] == ref.level(lvl)).all(), lvl
        off += cnt
    assert (cap == ref.cap).all()


@pytest.mark.parametrize("log_n,rate_bits,from_coeffs", [(18, 2, False), (18, 3, True), (19, 1, True), (18, 1, False), (15, 3, False),
                                                          (16, 2, False), (17, 3, False), (15, 1, True), (16, 3, True)])
def test_long_column_batches_at_other_blowups_match_the_oracle(ctx, log_n, rate_bits, from_coeffs):
    """the tree-of-rings transforms of ntt_tree.hip (N >= 2^15) through sipp_commit_batch_ex: blowup 2, 4 and 8 (2^rate_bits
    independent subtrees over the same coefficients; the fused middle sweep runs one after the other on its tile), three- and
    four-sweep plans (2^15 / 2^16 against 2^17 and above), from values and from coefficients -- coefficients, every LDE cell in leaf
    order and the cap equal the oracle's PolynomialBatch"""
    rng = np.random.default_rng(500 + log_n + rate_bits)
    ncols = 3 if log_n < 18 else 2
    data = _oracle.rand_field(rng, (ncols, 1 << log_n))
    ref = _oracle.Batch(data, log_n, rate_bits=rate_bits, cap_height=4, from_coeffs=from_coeffs)
    od, cap, (coeffs, lde, tree) = ctx.commit_ex(dev(data), log_n, rate_bits, 4, from_coeffs=from_coeffs)
    if not from_coeffs:
        assert (host(coeffs) == ref.coeffs).all()
    assert (host(lde).T == ref.leaves).all()
    assert (cap == ref.cap).all()


@pytest.mark.parametrize("log_n", [12, 16, 18])
def test_in_place_lde_takes_the_pass_by_pass_path(ctx, log_n):
    """values and coefficients in ONE buffer (sipp_lde_batch with d_coeffs == d_values): the fused kernels of a long column decline
    that (they read natural-order values while they write coefficients), so the call runs the pass-by-pass path -- copy, tiled
    bit-reversal, multi-pass DIT, multi-pass coset DIF -- at sizes the default commitments no longer send there.  Same
    coefficients and LDE cells as the oracle (and as the fused path: test_commit_matches_oracle)."""
    import torch
    rng = np.random.default_rng(900 + log_n)
    ncols = 3
    vals = _oracle.rand_field(rng, (ncols, 1 << log_n))
    ref = _oracle.Batch(vals, log_n)
    buf = dev(vals)
    lde = torch.empty((ncols, 2 << log_n), dtype=torch.int64, device=buf.device)
    ctx._ck(ctx.L.sipp_lde_batch(ctx.h, buf.data_ptr(), buf.data_ptr(), lde.data_ptr(), ncols, log_n), "lde_batch in place")
    assert (host(buf) == ref.coeffs).all()
    assert (host(lde).T == ref.leaves).all()


def test_leaves_and_cap_separately(ctx):
    rng = np.random.default_rng(5)
    log_n, ncols = 9, 20
    vals = _oracle.rand_field(rng, (ncols, 1 << log_n))
    ref = _oracle.Batch(vals, log_n)
    import torch
    coeffs, lde = ctx.lde(dev(vals), log_n)
    dig = ctx.poseidon_leaves(lde, log_n + 1)
    assert (host(dig) == ref.level(0)).all()
    tree = torch.zeros((4 << log_n, 4), dtype=torch.int64, device="cuda")
    tree[: 2 << log_n] = dig
    assert (ctx.merkle_cap(tree, log_n + 1) == ref.cap).all()


@pytest.mark.parametrize("log_leaves,ncols", [(5, 5), (5, 8), (6, 9), (7, 16), (9, 63), (12, 24), (14, 41), (16, 12)])
def test_thin_tree_leaf_hasher_matches_the_oracle(ctx, log_leaves, ncols):
    """the two-lanes-per-state leaf kernel (poseidon_pair.hpp: linear layers on the matrix pipe, lanes l / l + 32 share a leaf) on the
    launches it serves -- 32 .. 2^16 leaves, more than four columns, whole and ragged last chunks -- against hash_n_to_hash_no_pad of the
    oracle, leaf by leaf; the cells include 0, p - 1 and the values around 2^32 and 2^63"""
    rng = np.random.default_rng(1000 * log_leaves + ncols)
    n = 1 << log_leaves
    cells = _oracle.rand_field(rng, (ncols, n))
    edge = np.array([0, 1, P - 1, P - 2, (1 << 32) - 1, 1 << 32, (1 << 32) + 1, (1 << 63) - 1, 1 << 63, 0xFFFFFFFF00000000], dtype=np.uint64)
    cells[:, : len(edge)] = edge[None, :]
    cells[0, :] = P - 1
    dig = host(ctx.poseidon_leaves(dev(cells), log_leaves))
    step = max(1, n // 257)
    for j in list(range(0, 16)) + list(range(16, n, step)) + [n - 1]:
        assert (dig[j] == _oracle.hash_no_pad(cells[:, j])).all(), (log_leaves, ncols, j)


def test_hash_or_noop_narrow_leaves(ctx):
    rng = np.random.default_rng(6)
    for ncols in (1, 3, 4, 5, 8, 9):
        vals = _oracle.rand_field(rng, (ncols, 64))
        ref = _oracle.Batch(vals, 6)
        _, _, tree, cap = ctx.commit(dev(vals), 6)
        assert (host(tree)[:128] == ref.level(0)).all(), ncols
        assert (cap == ref.cap).all(), ncols
