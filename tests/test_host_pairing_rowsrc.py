"""Host tables of the final-pairing witness on the GPU (sipp_amd/csrc/pairing_rowsrc.h; API kind 6, reference
src/bin/bls_aggregation.rs:76-77): the symbolic replay of the row program says for every primary cell of the trace which value of the
record's pool it holds.  Checked here on the CPU against the oracle's trace: rebuilt from the oracle's OWN result column through the
table, every primary cell must come out the same; and the rows the value walk logs into are where tools/pairing_sched.py's
operations have their results."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import _oracle

ROOT = os.path.join(os.path.dirname(__file__), "..")
sys.path.insert(0, os.path.join(ROOT, "oracle", "py"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bn254 as bn  # noqa: E402
import pairing_rows as PR  # noqa: E402
import pairing_sched as PS  # noqa: E402


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("rowsrc") / "rowsrc.so")
    subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", "-Wall", "-Werror", "-I", os.path.join(ROOT, "data"), "-I", os.path.join(ROOT, "sipp_amd", "csrc"),
                           os.path.join(ROOT, "tests", "host", "pairing_rowsrc_shim.c"), "-o", so])
    return ctypes.CDLL(so)


def limbs(v):
    return [(v >> (16 * i)) & 0xFFFF for i in range(16)]


def test_trace_rebuilt_from_the_result_column(shim):
    import random
    rnd = random.Random(77)
    Pt, Q = bn.g1_mul(bn.G1, rnd.randrange(1, bn.R)), bn.g2_mul(bn.G2, rnd.randrange(1, bn.R))
    rec = np.array(bn.g1_to_u32(Pt) + bn.g2_to_u32(Q) + bn.f12_to_u32(bn.pairing(Pt, Q)), dtype=np.uint32)
    tr = _oracle.Trace(6, rec.reshape(1, 144))
    t = tr.array()
    R, E, N = shim.shim_rows(), shim.shim_elems(), shim.shim_pool()
    assert (R, E) == (8192, 147) and N < 65536
    src = np.zeros((E, R), dtype=np.uint16)
    assert shim.shim_sources(src.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16))) == 0
    res_col = shim.shim_elem_col(1, E - 1)
    assert res_col == tr.air.checked_base
    pool = np.zeros((N, 16), dtype=np.int64)
    pool[1], pool[2] = limbs(Pt[0]), limbs(Pt[1])
    for i, v in enumerate((Q[0][0], Q[0][1], Q[1][0], Q[1][1])):
        pool[3 + i] = limbs(v)
    for i, v in enumerate((bn.FROB_X[0], bn.FROB_X[1], bn.FROB_Y[0], bn.FROB_Y[1])):
        pool[7 + i] = limbs(v)
    g0 = shim.shim_pool_gc()
    for k, (u, v) in enumerate(PR.GC_PAIRS):
        pool[g0 + 2 * k], pool[g0 + 2 * k + 1] = limbs(u), limbs(v)
    r0 = shim.shim_pool_row()
    chk = t[res_col:res_col + 32, :R].astype(np.int64)
    pool[r0:r0 + R] = (chk[0::2] + 256 * chk[1::2]).T
    for e in range(E - 1):
        col = shim.shim_elem_col(1, e)
        assert (t[col:col + 16, :R].astype(np.int64) == pool[src[e]].T).all(), e
    # the result column itself: zero where the row has no result, the copied operand on the copy rows
    assert (pool[src[E - 1]] == pool[r0:r0 + R]).all()
    # every primary column is covered exactly once
    cols = sorted(c for e in range(E) for c in range(shim.shim_elem_col(1, e), shim.shim_elem_col(1, e) + (32 if e == E - 1 else 16)))
    reg_end = shim.shim_elem_col(1, E - 2) + 16       # (the gadget's sign cell sits between the registers and the checked cells)
    assert cols == list(range(1, reg_end)) + list(range(res_col, res_col + 32))


def test_logged_rows_hold_the_operations_results(shim):
    Pt, Q = bn.g1_mul(bn.G1, 5), bn.g2_mul(bn.G2, 9)
    ops, rows = [], []
    PS.simulate(Pt, Q, ops)
    PR.simulate_rows(Pt, Q, rows)
    oprow = np.zeros(512, dtype=np.int16)
    steprow = np.zeros(128, dtype=np.int16)
    ns = shim.shim_log_rows(oprow.ctypes.data_as(ctypes.POINTER(ctypes.c_int16)), steprow.ctypes.data_as(ctypes.POINTER(ctypes.c_int16)), 128)
    assert ns == 102
    s = 0
    for t, r in enumerate(PS.SCHEDULE):
        v = ops[t]
        if r["gop"] == PS.G_FQ:
            assert [rows[i]["res"] for i in range(8)] == [x for pr in v["S"][:4] for x in pr]
        elif r["gop"] != PS.G_IDLE:
            b = int(steprow[s])
            s += 1
            got = [rows[b + i]["res"] for i in (0, 1, 4, 5, 6, 7, 8, 9, 10, 11)]
            assert got == [x for pr in v["S"] for x in pr], t
        if oprow[t] >= 0:
            assert [rows[int(oprow[t]) + i]["res"] for i in range(12)] == [x for pr in v["C"] for x in pr], t
        else:
            assert r["fop"] == PS.F_IDLE or (r["fop"] == PS.F_LINE and t == 1)
    assert s == ns
