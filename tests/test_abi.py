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


def test_abi_version_and_checked_create():
    """a caller built against another header is refused before its (shorter) sipp_stark_config is read; the Rust shim's
    constant follows the header's"""
    import ctypes as C
    L = sipp_amd.lib()
    hdr = open(os.path.join(ROOT, "include", "sipp_hip.h")).read()
    ver = int(re.search(r"#define SIPP_ABI_VERSION (\d+)u", hdr).group(1))
    assert L.sipp_abi_version() == ver
    ffi = open(os.path.join(ROOT, "rust_shim", "src", "ffi.rs")).read()
    assert int(re.search(r"pub const SIPP_ABI_VERSION: u32 = (\d+);", ffi).group(1)) == ver
    assert "sipp_ctx_create_checked" in open(os.path.join(ROOT, "rust_shim", "src", "lib.rs")).read()
    cfg = sipp_amd.default_config()
    h = C.c_void_p(1)
    assert L.sipp_ctx_create_checked(C.byref(h), 0, C.byref(cfg), C.sizeof(cfg), ver + 1, 1 << 20) == -1 and not h.value   # SIPP_E_BADARG
    h = C.c_void_p(1)
    assert L.sipp_ctx_create_checked(C.byref(h), 0, C.byref(cfg), C.sizeof(cfg) - 8, ver, 1 << 20) == -1 and not h.value  # the round-4 struct


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


# ---- rust_shim/src/ffi.rs against include/sipp_hip.h (the shim cannot be compiled here: no Rust toolchain) ----------------------
_STRUCTS = {"sipp_ctx": "SippCtxOpaque", "sipp_stark_config": "SippStarkConfig", "sipp_fri_params": "SippFriParams",
            "sipp_oracle": "SippOracle", "sipp_poly_range": "SippPolyRange", "sipp_fri_batch": "SippFriBatch",
            "sipp_challenger": "SippChallenger", "sipp_plonk_params": "SippPlonkParams", "sipp_plonk_gate": "SippPlonkGate",
            "sipp_plonk_circuit": "SippPlonkCircuit", "sipp_plonk_generator": "SippPlonkGenerator", "sipp_plonk_schedule": "SippPlonkSchedule", "sipp_plonk_schedule_host": "SippPlonkScheduleHost",
            "sipp_circuit_data": "SippCircuitDataOpaque"}
_SCALARS = {"int": "c_int", "size_t": "usize", "uint32_t": "u32", "uint64_t": "u64", "int64_t": "i64", "float": "f32", "char": "c_char", "void": "c_void"}


def _c_to_rust(ctype):
    """one C parameter / field / return type (declarator stripped of its name) -> the Rust spelling ffi.rs must use"""
    t = ctype.strip()
    arr = re.search(r"\[\s*(\w*)\s*\]$", t)          # parameter arrays decay to pointers
    if arr:
        t = t[: arr.start()].strip() + " *"
    toks = re.findall(r"\*|\w+", t)
    base_const = False
    base = None
    i = 0
    while i < len(toks) and toks[i] != "*":
        if toks[i] == "const":
            base_const = True
        else:
            base = toks[i]
        i += 1
    rust = _SCALARS.get(base) or _STRUCTS[base]
    cur_const = base_const
    while i < len(toks):                               # every '*' wraps what is left of it; a 'const' after it qualifies the pointer
        assert toks[i] == "*"
        rust = ("*const " if cur_const else "*mut ") + rust
        cur_const = False
        i += 1
        while i < len(toks) and toks[i] == "const":
            cur_const = True
            i += 1
    if rust == "c_void":
        return "()"
    return rust


def _split_decl(decl):
    """'const uint32_t *const ios[3]' -> (type, name)"""
    decl = decl.strip()
    m = re.match(r"^(.*?)(\w+)\s*(\[\s*\w*\s*\])?$", decl, flags=re.S)
    return (m.group(1) + (m.group(3) or "")).strip(), m.group(2)


def _header_api():
    txt = open(os.path.join(ROOT, "include", "sipp_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = "\n".join(l for l in txt.splitlines() if not l.strip().startswith("#"))
    structs = {}
    for body, name in re.findall(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", txt, flags=re.S):
        fields = []
        for stmt in body.split(";"):
            stmt = stmt.strip()
            if not stmt:
                continue
            first_type, first_name = _split_decl(stmt.split(",")[0])
            base = re.sub(r"\[.*\]$", "", first_type).strip()
            for part in stmt.split(","):
                part = part.strip()
                t, n = (first_type, first_name) if part == stmt.split(",")[0].strip() else _split_decl(base + " " + part)
                dim = re.search(r"\[\s*(\w+)\s*\]$", t)
                if dim:
                    fields.append((n, "[%s; %s]" % (_c_to_rust(t[: dim.start()]), dim.group(1))))
                else:
                    fields.append((n, _c_to_rust(t)))
        structs[name] = fields
    txt = re.sub(r"typedef\s+(struct|enum)\s*\{.*?\}\s*\w+\s*;", "", txt, flags=re.S)
    txt = re.sub(r"typedef\s+struct\s+\w+\s+\w+\s*;", "", txt)
    funcs = {}
    for ret, name, args in re.findall(r"([\w\s\*]+?)\b(sipp_\w+)\s*\(([^)]*)\)\s*;", txt, flags=re.S):
        params = [] if args.strip() in ("", "void") else [_split_decl(a) for a in args.split(",")]
        funcs[name] = ([(n, _c_to_rust(t)) for t, n in params], _c_to_rust(ret))
    return structs, funcs


def _rust_api():
    txt = open(os.path.join(ROOT, "rust_shim", "src", "ffi.rs")).read()
    txt = re.sub(r"//[^\n]*", "", txt)
    structs = {}
    for name, body in re.findall(r"pub struct (\w+)\s*\{(.*?)\}", txt, flags=re.S):
        structs[name] = [(n, " ".join(t.split())) for n, t in re.findall(r"pub\s+(\w+)\s*:\s*([^,\n]+),", body)]
    ext = re.search(r'extern "C"\s*\{(.*)\}', txt, flags=re.S).group(1)
    funcs = {}
    for name, args, ret in re.findall(r"pub fn (\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", ext, flags=re.S):
        params = []
        for a in [x for x in args.split(",") if x.strip()]:
            n, t = a.split(":", 1)
            params.append((n.strip().replace("r#", ""), " ".join(t.split())))
        funcs[name] = (params, " ".join(ret.split()) if ret else "()")
    return structs, funcs


def test_rust_bindings_match_the_header():
    """every function of include/sipp_hip.h is bound in rust_shim/src/ffi.rs with the same parameter names, order and types and
    the same return type; every struct has the same fields in the same order; nothing is bound that the header does not declare"""
    c_structs, c_funcs = _header_api()
    r_structs, r_funcs = _rust_api()
    assert sorted(c_funcs) == declared_symbols()
    assert sorted(r_funcs) == sorted(c_funcs), (sorted(set(c_funcs) - set(r_funcs)), sorted(set(r_funcs) - set(c_funcs)))
    for name, (params, ret) in c_funcs.items():
        assert r_funcs[name] == (params, ret), (name, r_funcs[name], (params, ret))
    assert len(c_structs) == 12
    for cname, fields in c_structs.items():
        rfields = r_structs[_STRUCTS[cname]]
        want = [(n, t.replace("SIPP_FRI_MAX_ROUNDS", "SIPP_FRI_MAX_ROUNDS")) for n, t in fields]
        assert rfields == want, (cname, rfields, want)
    # the shim's safe layer calls only what ffi.rs binds
    lib_rs = open(os.path.join(ROOT, "rust_shim", "src", "lib.rs")).read()
    used = set(re.findall(r"ffi::(sipp_\w+)", lib_rs))
    assert used and used <= set(r_funcs), used - set(r_funcs)
