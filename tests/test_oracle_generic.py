"""Pins the oracle's generic layer: Poseidon against the three upstream permutation KATs
(SURVEY.md App. E), the sponge/compression vectors derived there, and the NTT against the
naive DFT / polynomial evaluation identities.  CPU only."""
import numpy as np

from tests import _oracle
from tests._oracle import P

KAT = [
    ([0] * 12,
     [0x3c18a9786cb0b359, 0xc4055e3364a246c3, 0x7953db0ab48808f4, 0xc71603f33a1144ca,
      0xd7709673896996dc, 0x46a84e87642f44ed, 0xd032648251ee0b3c, 0x1c687363b207df62,
      0xdf8565563e8045fe, 0x40f5b37ff4254dae, 0xd070f637b431067c, 0x1792b1c4342109d7]),
    (list(range(12)),
     [0xd64e1e3efc5b8e9e, 0x53666633020aaa47, 0xd40285597c6a8825, 0x613a4f81e81231d2,
      0x414754bfebd051f0, 0xcb1f8980294a023f, 0x6eb2a9e4d54a9d0f, 0x1902bc3af467e056,
      0xf045d5eafdc6021f, 0xe4150f77caaa3be5, 0xc9bfd01d39b50cce, 0x5c0a27fcb0e1459b]),
    ([P - 1] * 12,
     [0xbe0085cfc57a8357, 0xd95af71847d05c09, 0xcf55a13d33c1c953, 0x95803a74f4530e82,
      0xfcd99eb30a135df1, 0xe095905e913a3029, 0xde0392461b42919b, 0x7d3260e24e81d031,
      0x10d3d0465d9deaa0, 0xa87571083dfc2a47, 0xe18263681e9958f8, 0xe28e96f1ae5e60d3]),
]


def test_poseidon_permutation_kats():
    for inp, out in KAT:
        assert [int(x) for x in _oracle.permute(inp)] == out


def test_sponge_vectors():
    h = lambda v: [int(x) for x in _oracle.hash_no_pad(v)]
    assert h([1]) == [0xd074b8cee5dcf415, 0x2346a1b4c0f390e8, 0x47969c1f5a6a25b1, 0xda62fdf84a21108e]
    assert h([0]) == KAT[0][1][:4]
    assert h(list(range(8))) == [0xeff81bb29a227619, 0x7ec080e2b7f39736, 0xf624fcbf98c9e736, 0xc4221df46aa44e4c]
    assert h(list(range(9))) == [0xf9e711e9767ee486, 0x98cd7988e37bdad8, 0x0397e8ac2fd0408d, 0x7d8cf7363df72353]
    assert h(list(range(20))) == [0xf9fa02631df4a49f, 0x1d4fcc61eaa20c62, 0x8c6bd13cd03741e7, 0xe01ba2d37d5a4adf]
    assert h(list(range(100))) == [0xbaadcd55e7879422, 0x85e5f82c91f46067, 0x7cd4841ef2261a00, 0x52fecb7bbfd661bd]
    t = [int(x) for x in _oracle.two_to_one([1, 2, 3, 4], [5, 6, 7, 8])]
    assert t == [0xd110aa6a46373941, 0x8f238fcceb658894, 0x9cd4f8353866fb4f, 0x274913f0007aa232]
    assert [int(x) for x in _oracle.two_to_one([0] * 4, [0] * 4)] == KAT[0][1][:4]


def test_fft_matches_naive_dft(oracle):
    rng = np.random.default_rng(7)
    for log_n in (1, 2, 5, 8):
        a = _oracle.rand_field(rng, 1 << log_n)
        ref = np.zeros_like(a)
        oracle.orc_naive_dft(a, ref, log_n)
        b = a.copy()
        oracle.orc_fft(b, log_n)
        assert (b == ref).all()
        oracle.orc_ifft(b, log_n)
        assert (b == a).all()


def test_coset_lde_is_polynomial_evaluation(oracle):
    rng = np.random.default_rng(8)
    log_n = 5
    n = 1 << log_n
    c = _oracle.rand_field(rng, n)
    out = np.zeros(2 * n, dtype=np.uint64)
    oracle.orc_coset_lde(c, log_n, 1, 7, out)
    w = pow(1753635133440165772, 1 << (32 - log_n - 1), P)
    for i in (0, 1, 17, 2 * n - 1):
        x = 7 * pow(w, i, P) % P
        assert int(out[i]) == sum(int(cj) * pow(x, j, P) for j, cj in enumerate(c)) % P


def test_batch_commit_structure():
    rng = np.random.default_rng(9)
    log_n, ncols = 6, 11
    vals = _oracle.rand_field(rng, (ncols, 1 << log_n))
    b = _oracle.Batch(vals, log_n)
    leaves = b.leaves
    m = 2 << log_n
    # leaf j = natural LDE row bitrev(j); check via direct evaluation of column 3 at two leaves
    coeffs = b.coeffs
    w = pow(1753635133440165772, 1 << (32 - log_n - 1), P)
    br = lambda x, bits: int(format(x, "0%db" % bits)[::-1], 2)
    for j in (1, 77):
        x = 7 * pow(w, br(j, log_n + 1), P) % P
        assert int(leaves[j, 3]) == sum(int(cj) * pow(x, k, P) for k, cj in enumerate(coeffs[3])) % P
    # digests: level 0 = hash_no_pad(leaf), level l+1 = two_to_one
    l0 = b.level(0)
    assert (l0[5] == _oracle.hash_no_pad(leaves[5])).all()
    l1 = b.level(1)
    assert (l1[2] == _oracle.two_to_one(l0[4], l0[5])).all()
    assert b.cap.shape == (16, 4)
    assert (b.cap == b.level(log_n + 1 - 4)).all()
    assert m == leaves.shape[0]
