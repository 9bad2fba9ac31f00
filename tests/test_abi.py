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
