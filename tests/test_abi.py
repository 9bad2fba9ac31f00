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
