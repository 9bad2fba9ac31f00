"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/sipp_hip.h declares (no compute calls without a GPU)."""
import os
import re

import sipp_amd
from sipp_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "sipp_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sipp_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_all_exported():
    L = sipp_amd.lib()
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(L, s), "libsipp_hip.so does not export %s" % s


def test_binding_covers_header():
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_default_config_is_standard_fast_config():
    cfg = sipp_amd.default_config()
    assert (cfg.rate_bits, cfg.cap_height, cfg.pow_bits, cfg.arity_bits, cfg.final_poly_bits, cfg.num_queries,
            cfg.num_challenges) == (1, 4, 16, 4, 5, 84, 2)


def test_no_gpu_fails_loudly():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(sipp_amd.SippError):
        sipp_amd.Ctx()


def test_io_shard_ranges_tile_the_list_and_reject_bad_ranks():
    """sipp_io_shard (IO-sharded sub-proofs, DESIGN.md section 5 level L-D) is host arithmetic: callable without a GPU"""
    import pytest
    for n in (0, 1, 6, 7, 14, 127, 1023, 4095):
        for world in (1, 2, 3, 4, 8):
            nxt = 0
            for rank in range(world):
                first, count = sipp_amd.io_shard(n, world, rank)
                assert first == nxt and n // world <= count <= -(-n // world)
                nxt = first + count
            assert nxt == n
    with pytest.raises(sipp_amd.SippError):
        sipp_amd.io_shard(10, 4, 4)
    with pytest.raises(sipp_amd.SippError):
        sipp_amd.io_shard(10, 0, 0)


def test_host_poseidon_implementations_agree_with_the_oracle():
    """The Fiat-Shamir permutation on the host (sipp_amd/csrc/host_poseidon.cpp): every implementation the CPU supports --
    portable scalar, look-ahead partial rounds, AVX-512 -- equals the CPU oracle's naive permutation, on canonical words,
    on non-canonical words (any u64 is a legal input) and on the corner values; -1 = the one the provers use here."""
    import ctypes as C
    import numpy as np
    from tests import _oracle
    L = sipp_amd.lib()
    rng = np.random.default_rng(20260104)
    states = _oracle.rand_field(rng, (600, 12))
    states[0] = 0
    states[1] = _oracle.P - 1
    states[2, ::2] = 0
    states[3] = np.arange(12, dtype=np.uint64)
    want = np.stack([_oracle.permute(st) for st in states])
    # KAT of the permutation: the all-zero state
    assert int(want[0][0]) == 0x3c18a9786cb0b359
    ran = 0
    for impl in (-1, 0, 1, 2, 3):
        got = states.copy()
        rc = L.sipp_host_poseidon_permute(got.ctypes.data, got.shape[0], impl)
        if rc != 0:
            assert impl >= 2, "only the AVX-512 forms may be unavailable"
            continue
        assert (got == want).all(), impl
        ran += 1
    assert ran >= 3
    # non-canonical inputs: the same words plus p (where that fits 64 bits) must give the same result
    small = states < (1 << 32) - 1
    noncanon = np.where(small, states + np.uint64(_oracle.P), states)
    assert (noncanon != states).any()
    for impl in (-1, 0, 1, 2, 3):
        got = noncanon.copy()
        if L.sipp_host_poseidon_permute(got.ctypes.data, got.shape[0], impl) == 0:
            assert (got == want).all(), impl
    assert L.sipp_host_poseidon_permute(states.ctypes.data, 1, 7) != 0
