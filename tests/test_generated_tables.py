"""The generated constant tables of the hash kernels (tools/gen_poseidon_header.py, tools/gen_gl_muln.py): the committed files are what
the generators emit, and the Python models of the matrix-pipe forms -- the operand layout of v_mfma_i32_32x32x32_i8 with one state per
lane (poseidon.hpp::dense_mfma) and with two lanes per state (poseidon_pair.hpp: two byte planes and two digits per instruction) --
reproduce plain matrix-vector products mod p on arbitrary u64 inputs.  (The kernels themselves are held against the oracle on the GPU:
tests/test_gpu_generic.py.)"""
import os
import random
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _regenerates_identically(script, outputs, tmp_path):
    """the generator writes below tmp_path (--out-root); the tracked sources are only READ (their mtimes must not move: sipp_amd/build.py
    would relink the library under a running test process)"""
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", script), "--out-root", str(tmp_path)], stdout=subprocess.DEVNULL)
    for p in outputs:
        assert open(os.path.join(str(tmp_path), p), "rb").read() == open(os.path.join(ROOT, p), "rb").read(), \
            "%s is not what tools/%s emits" % (p, script)


def test_poseidon_headers_are_current(tmp_path):
    # (the generator also re-derives the fast partial-round tables and checks every form against the naive permutation before it writes)
    _regenerates_identically("gen_poseidon_header.py", ["sipp_amd/csrc/poseidon_constants.h", "oracle/poseidon_constants.h"], tmp_path)


def test_interleaved_product_block_is_current(tmp_path):
    _regenerates_identically("gen_gl_muln.py", ["sipp_amd/csrc/gl_lazy_muln.inc"], tmp_path)


def test_matrix_pipe_models_reproduce_matrix_products():
    import gen_poseidon_header as g
    P = g.P
    rc = g.load_rc()
    first, scalars = g.derive_fast_partial(rc)
    Mi, vs, ws = g.sparse_factor()
    rnd = random.Random(4)
    xs = [[0] * 12, [(1 << 64) - 1] * 12, [P - 1] * 12] + [[rnd.randrange(1 << 64) for _ in range(12)] for _ in range(3)]
    # one state per lane: five dense products
    mats = g.dense_matrices(first, Mi, vs, ws)
    frag, starts = g.dense_tables(mats)
    for mi, (Mx, add) in enumerate(mats):
        for x in xs:
            extra = [rnd.randrange(1 << 64) for _ in range(12)] if (mi and mi % 2 == 0) else None
            got = g.dense_model(mi, frag, starts, x, extra)
            for r in range(len(Mx)):
                assert got[r] == (sum(Mx[r][e] * x[e] for e in range(12)) + add[r] + (extra[r] if extra else 0)) % P, (mi, r)
    # two lanes per state: the MDS layer and the five dense products, through a model of the instruction's operand layout
    pmats = g.pair_matrices(first, Mi, vs, ws, rc)
    pfrag, pstarts = g.pair_tables(pmats)
    pmds = g.pair_mds_fragment()
    Mm = g.mds_matrix()
    for x in xs:
        addc = [rnd.randrange(P) for _ in range(12)]
        assert g.pair_mds_model(pmds, x, addc) == [(sum(Mm[r][e] * x[e] for e in range(12)) + addc[r]) % P for r in range(12)]
        for mi, (Mx, add) in enumerate(pmats):
            extra = [rnd.randrange(1 << 64) for _ in range(12)] if (mi and mi % 2 == 0) else None
            got = g.pair_dense_model(mi, pfrag, pstarts, x, extra)
            assert got == [(sum(Mx[r][e] * x[e] for e in range(12)) + add[r] + (extra[r] if extra else 0)) % P for r in range(12)], mi
    # the last V product carries the constants of full round 26 (elements 1 .. 11)
    assert pmats[4][1][1:] == [rc[12 * 26 + e] for e in range(1, 12)] and pmats[2][1] == [0] * 12


def test_pair_fragment_layout_follows_the_instruction():
    """row rho of A belongs to lane half (rho >> 2) & 1, register (rho & 3) + 4 (rho >> 3): registers 0 .. 5 = the lane's six outputs with
    the first digit / plane, 6 .. 11 with the second; bytes 0 .. 5 and 8 .. 13 of a lane's sixteen are its six inputs' two slots"""
    import gen_poseidon_header as g
    seen = {}     # (output half, register, input half) -> a lane holds it
    frag = g.pair_fragment(lambda eo, ds, ei, ps: (eo + 2 * ei + ds - ps) % 100)
    assert len(frag) == 64 * 4
    used = 0
    for lane in range(64):
        rho, hh = lane & 31, lane >> 5
        h_out, reg = g.pair_row(rho)
        words = frag[4 * lane:4 * lane + 4]
        by = [(words[b >> 2] >> (8 * (b & 3))) & 0xFF for b in range(16)]
        if reg >= 12:
            assert by == [0] * 16
            continue
        used += 1
        j, ds = reg % 6, reg // 6
        for b in range(16):
            ps, jj = b >> 3, b & 7
            want = (6 * h_out + j + 2 * (6 * hh + jj) + ds - ps) % 100 if jj < 6 else 0
            assert by[b] == want, (lane, b)
        seen[(h_out, reg, hh)] = True
    assert used == 48 and len(seen) == 48      # 2 output halves x 12 registers x 2 input halves


def test_committed_digest_files_cover_the_baseline_configs():
    """the sha256 digests the GPU parity tests compare with exist for every configuration they claim: n = 4 and n = 128 (plain, hardened,
    upstream rules), BASELINE configs[2] (n = 1024) and rank 3's world-8 shard of configs[4] (n = 4096), each with the oracle's shape"""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = json.load(open(os.path.join(root, "tests", "golden", "proof_digests_n128.json")))
    assert set(g) >= {"g1", "g2", "fq12", "g1_hardened", "g2_hardened", "g1_upstream_rules", "g2_upstream_rules", "fq12_upstream_rules"}
    big = json.load(open(os.path.join(root, "tests", "golden", "proof_digests_large.json")))
    for cfg, log_n, recs in (("n1024", 19, 1023), ("n4096_world8_rank3", 18, 512)):
        for key, kind in (("g1_hardened", 4), ("g2_hardened", 5), ("fq12", 2)):
            e = big["%s.%s" % (cfg, key)]
            assert e["kind"] == kind and len(e["sha256"]) == 64
            if kind != 2:
                assert e["log_n"] == log_n and e["records"] == recs
