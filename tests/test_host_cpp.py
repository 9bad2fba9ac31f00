"""The C++ host layer (include/sipp_host.hpp: the reference's G1ExpInput / g1_exp_circuit / StarkProofWithPublicInputs
names over the C ABI) through tests/host/test_sipp_circuit.cpp, which follows the reference's own test_sipp_circuit
(src/verifier_circuit.rs:192-269): obligation lists in, outputs + three verified proofs out."""
import os
import subprocess

import numpy as np
import pytest

from tests import _oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "tests", "host")
EXE = os.path.join(HOST, "test_sipp_circuit")


def build_host_test():
    import sipp_amd
    sipp_amd.lib()            # fails loudly when libsipp_hip.so is not built
    _oracle.build()
    subprocess.check_call(["make", "-C", HOST, "-s"])
    return EXE


def test_host_test_mirrors_the_oracle_config_struct():
    """tests/host/test_sipp_circuit.cpp declares the checker's orc_config itself (it links liboracle.so without its headers): the
    declaration must list the fields of oracle/stark.h in order -- a shorter mirror lets orc_default_config write past the object"""
    import re

    def fields(text, start):
        body = text[text.index(start):]
        body = body[body.index("{") + 1:body.index("}")]
        body = re.sub(r"/\*.*?\*/|//[^\n]*", "", body, flags=re.S)
        return [n.strip() for part in body.split(";") for n in part.replace("uint32_t", "").split(",") if n.strip()]
    want = fields(open(os.path.join(ROOT, "oracle", "stark.h")).read(), "typedef struct {")
    got = fields(open(os.path.join(HOST, "test_sipp_circuit.cpp")).read(), "struct orc_config {")
    assert want == got and len(want) == 10, (want, got)


def test_header_is_plain_c():
    """include/sipp_hip.h is the FFI surface: it must compile as C (what cgo / bindgen / ctypes consume)."""
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c",
                           os.path.join(ROOT, "include", "sipp_hip.h")])


def test_record_layouts_and_flat_roundtrip(tmp_path):
    exe = build_host_test()
    ios = np.load(os.path.join(ROOT, "tests", "golden", "sipp_n4_ios.npz"))["g1"]
    flat = _oracle.stark_prove(0, ios)
    path = tmp_path / "proof.bin"
    flat.tofile(path)
    out = subprocess.run([exe, "layout", str(path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "layout ok" in out.stdout
    # the parser on damaged buffers (header fields at extreme values, flipped words, truncations, extensions): every mutation
    # ends in a sipp::Error or in a parse that re-serialises to the same words (scripts/run_asan.sh runs 20,000 under ASan / UBSan)
    out = subprocess.run([exe, "fuzz", str(path), "4000"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "fuzz ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_sipp_circuit_cpp(tmp_path):
    """outputs from the device equal the native chain's, the three proofs verify, and they are word for word the proofs
    the Python harness gets from the same library"""
    import sipp_amd
    exe = build_host_test()
    d = np.load(os.path.join(ROOT, "tests", "golden", "sipp_n4_ios.npz"))
    ios = [np.ascontiguousarray(d[k], dtype=np.uint32) for k in ("g1", "g2", "fq12")]
    st = d["statement"]
    pts = [np.ascontiguousarray(st[:64].reshape(4, 16)), np.ascontiguousarray(st[64:192].reshape(4, 32))]   # A, B of the n = 4 fixture
    with open(tmp_path / "ios.bin", "wb") as f:
        for a in ios + pts:
            f.write(np.uint64(a.shape[0]).tobytes())
            f.write(a.tobytes())
            f.write(b"\0" * (-a.nbytes % 8))
    out = subprocess.run([exe, "prove", str(tmp_path / "ios.bin"), str(tmp_path / "proof")], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "native chain ok" in out.stdout
    assert "final pairing ok" in out.stdout          # pairing_circuit(final_A, final_B) == final_Z (src/bin/bls_aggregation.rs:76-77)
    ctx = sipp_amd.Ctx(workspace_bytes=8 << 30)
    try:
        for k in range(3):
            got = np.fromfile(tmp_path / ("proof%d.bin" % k), dtype=np.uint64)
            want = ctx.prove(k, ios[k])
            assert len(got) == len(want) and (got == want).all(), k
    finally:
        ctx.close()


@pytest.mark.gpu
def test_bls_front_end_cpp(tmp_path):
    """the BLS example's messages -> G2 step through the C++ host layer (src/bin/bls_aggregation.rs:65, :100-104):
    batch_map_to_g2_circuit (proof accepted by the oracle's verifier, equal to the Python harness's), map_to_g2 (cleared points equal
    the Python reading's), the cofactor obligations through g2_exp_circuit"""
    import sys
    import sipp_amd
    sys.path.insert(0, os.path.join(ROOT, "oracle", "py"))
    import bn254
    import map_to_g2 as M
    exe = build_host_test()
    us = [(5, 7), (0, 3), (bn254.P - 2, 11)]
    words = np.array([bn254.fq_to_u32(u[0]) + bn254.fq_to_u32(u[1]) for u in us], dtype=np.uint32)
    with open(tmp_path / "msgs.bin", "wb") as f:
        f.write(np.uint64(len(us)).tobytes())
        f.write(words.tobytes())
    out = subprocess.run([exe, "mapg2", str(tmp_path / "msgs.bin"), str(tmp_path / "m")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "mapg2 ok" in out.stdout, out.stdout + out.stderr
    pts = np.fromfile(tmp_path / "m_points.bin", dtype=np.uint32).reshape(len(us), 32)
    for u, p in zip(us, pts):
        assert list(p) == bn254.g2_to_u32(M.map_to_g2(u))
    got = np.fromfile(tmp_path / "m_proof.bin", dtype=np.uint64)
    ctx = sipp_amd.Ctx(workspace_bytes=2 << 30)
    try:
        want = ctx.prove(3, ctx.map_to_g2(words, cofactor=False))
    finally:
        ctx.close()
    assert len(got) == len(want) and (got == want).all()


@pytest.mark.gpu
def test_outer_circuit_data_cpp(tmp_path):
    """sipp::CircuitData (include/sipp_host.hpp over sipp_circuit_build / _prove / _verify): the tail of the reference's test --
    builder.build, data.prove(pw), data.verify(proof), public inputs read back as the SIPPStatement (src/verifier_circuit.rs:225, :253-268)
    -- in C++ with host memory only, on the chained recursion-shaped stand-in circuit with the n = 4 fixture's statement as public
    inputs; the proof the binary writes is the ORACLE's proof of the full numpy witness word for word"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import plonk_synth as ps
    from tests import _oracle
    from tests.test_oracle_plonk import fri
    exe = build_host_test()
    st = np.load(os.path.join(ROOT, "tests", "golden", "sipp_n4_ios.npz"))["statement"].astype(np.uint64)
    log_n, chain = 10, 8
    circ = ps.circuit(136, 80)
    pih = [int(x) for x in _oracle.hash_no_pad(st)]
    wires, cs, gate = ps.witness(circ, log_n, 17, pih, chain_len=chain)
    w_in = ps.witness(circ, log_n, 17, pih, inputs_only=True, chain_len=chain)[0]
    sc = ps.chain_schedule(log_n, chain)
    ofp = fri(log_n, rate_bits=3, cap_height=4, nq=28, arity=4, fpb=5)
    gens = ps.generators(circ)
    u64 = lambda v: np.ascontiguousarray(v, dtype=np.uint64).reshape(-1)
    head = [log_n, circ["num_wires"], circ["num_routed"], circ["num_constants"], circ["num_selectors"], len(circ["gates"]), len(circ["programs"]),
            len(gens), sc["n_levels"], len(sc["copy_src"]), len(st), 8]
    fp_words = [ofp.rate_bits, ofp.cap_height, ofp.pow_bits, ofp.num_queries, ofp.pow_rule, ofp.hiding, ofp.n_rounds] + [int(x) for x in ofp.arity_bits]
    parts = [u64(head), u64(fp_words), u64([list(g) for g in circ["gates"]]), circ["programs"].astype(np.int64).view(np.uint64), u64([list(g) for g in gens]),
             u64(sc["rows"]), u64(sc["level_offsets"]), u64(sc["copy_src"]), u64(sc["copy_dst"]), u64(sc["copy_offsets"]), u64(cs), u64(w_in), u64(st)]
    with open(tmp_path / "circuit.bin", "wb") as f:
        for a in parts:
            f.write(a.tobytes())
    out = subprocess.run([exe, "outer", str(tmp_path / "circuit.bin"), str(tmp_path / "outer.bin")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "outer proof ok" in out.stdout, out.stdout + out.stderr
    got = np.fromfile(tmp_path / "outer.bin", dtype=np.uint64)
    pf, digest = got[:-4], [int(x) for x in got[-4:]]
    want = _oracle.plonk_prove_gates(wires, cs, log_n, _oracle.plonk_params(80, 8, 2), ofp, circ, digest, [int(x) for x in st])
    assert len(pf) == len(want) and (pf == want).all()
