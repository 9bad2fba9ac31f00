"""ctypes binding of include/sipp_hip.h -- the same surface a Rust `extern "C"` block would bind
(INTEGRATION.md).  Device memory is plain torch tensors (dtype int64 viewed as u64)."""
import ctypes as C
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libsipp_hip.so")

P = 2**64 - 2**32 + 1

STATUS = {
    0: "SIPP_OK", -1: "SIPP_E_BADARG", -2: "SIPP_E_HIP", -3: "SIPP_E_NOMEM", -4: "SIPP_E_BUFSZ",
    -5: "SIPP_E_SUBGROUP", -6: "SIPP_E_QUOTIENT", -7: "SIPP_E_UNSUPPORTED", -8: "SIPP_E_WITNESS", -9: "SIPP_E_VERIFY",
}


class SippError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__("%s (%d): %s" % (STATUS.get(code, "?"), code, msg))
        self.code = code


class StarkConfig(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("rate_bits", "cap_height", "pow_bits", "arity_bits", "final_poly_bits",
                                          "num_queries", "num_challenges", "pow_rule", "fs_rule", "lookup_rule")]


class FriParams(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("rate_bits", "cap_height", "pow_bits", "num_queries", "pow_rule", "hiding", "n_rounds")] + \
               [("arity_bits", C.c_uint32 * 32)]


class Oracle(C.Structure):
    _fields_ = [("d_coeffs", C.c_void_p), ("d_lde", C.c_void_p), ("d_tree", C.c_void_p), ("n_polys", C.c_uint32), ("n_salt", C.c_uint32)]


class PolyRange(C.Structure):
    _fields_ = [("oracle", C.c_uint32), ("col_begin", C.c_uint32), ("col_end", C.c_uint32)]


class FriBatch(C.Structure):
    _fields_ = [("point", C.c_uint64 * 2), ("n_ranges", C.c_uint32), ("ranges", C.POINTER(PolyRange))]


class PlonkParams(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("num_routed_wires", "max_degree", "num_challenges")]


class PlonkGate(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("selector_index", "row", "group_lo", "group_hi", "prog_offset", "num_constraints")]


class PlonkCircuit(C.Structure):
    """sipp_plonk_circuit: the gate set as data (include/sipp_hip.h, "gates as data")"""
    _fields_ = [("num_wires", C.c_uint32), ("num_constants", C.c_uint32), ("num_selectors", C.c_uint32), ("num_gates", C.c_uint32),
                ("gates", C.POINTER(PlonkGate)), ("programs", C.POINTER(C.c_int64)), ("program_words", C.c_uint32)]

    @classmethod
    def from_dict(cls, circ):
        """circ: dict(num_wires, num_constants, num_selectors, gates=[(selector_index, row, lo, hi, prog_offset, n_constraints)], programs)"""
        gates = (PlonkGate * len(circ["gates"]))(*[PlonkGate(*[int(x) for x in g]) for g in circ["gates"]])
        prog = np.ascontiguousarray(circ["programs"], dtype=np.int64)
        c = cls(circ["num_wires"], circ["num_constants"], circ["num_selectors"], len(circ["gates"]), gates,
                prog.ctypes.data_as(C.POINTER(C.c_int64)), len(prog))
        c._keep = (gates, prog)           # the struct points into these
        return c


class PlonkGenerator(C.Structure):
    """sipp_plonk_generator: one gate family's witness generator with its layout (include/sipp_hip.h, "WITNESS GENERATORS")"""
    _fields_ = [("kind", C.c_uint32), ("selector_index", C.c_uint32), ("row", C.c_uint32), ("p", C.c_uint32 * 5)]


class PlonkSchedule(C.Structure):
    """sipp_plonk_schedule: rows sorted by level + the copies each level's outputs feed (include/sipp_hip.h)"""
    _fields_ = [("n_levels", C.c_uint32), ("d_rows", C.c_void_p), ("level_offsets", C.POINTER(C.c_uint32)), ("d_copy_src", C.c_void_p),
                ("d_copy_dst", C.c_void_p), ("copy_offsets", C.POINTER(C.c_uint32))]

    @classmethod
    def from_dict(cls, sched):
        """sched: tools/plonk_synth.chain_schedule() -- uploads rows / copy cells, keeps the host offsets alive"""
        import torch
        rows = torch.from_numpy(np.ascontiguousarray(sched["rows"], dtype=np.uint32).view(np.int32)).cuda()
        src = to_device(np.ascontiguousarray(sched["copy_src"] if len(sched["copy_src"]) else [0], dtype=np.uint64))
        dst = to_device(np.ascontiguousarray(sched["copy_dst"] if len(sched["copy_dst"]) else [0], dtype=np.uint64))
        lo = np.ascontiguousarray(sched["level_offsets"], dtype=np.uint32)
        co = np.ascontiguousarray(sched["copy_offsets"], dtype=np.uint32)
        s = cls(int(sched["n_levels"]), rows.data_ptr(), lo.ctypes.data_as(C.POINTER(C.c_uint32)), src.data_ptr(), dst.data_ptr(),
                co.ctypes.data_as(C.POINTER(C.c_uint32)))
        s._keep = (rows, src, dst, lo, co)
        return s


class PlonkScheduleHost(C.Structure):
    """sipp_plonk_schedule_host: the level schedule as host arrays (sipp_circuit_build copies them)"""
    _fields_ = [("n_levels", C.c_uint32), ("rows", C.POINTER(C.c_uint32)), ("level_offsets", C.POINTER(C.c_uint32)), ("copy_src", C.POINTER(C.c_uint64)),
                ("copy_dst", C.POINTER(C.c_uint64)), ("copy_offsets", C.POINTER(C.c_uint32))]

    @classmethod
    def from_dict(cls, sched):
        a32 = lambda v: np.ascontiguousarray(v, dtype=np.uint32)
        a64 = lambda v: np.ascontiguousarray(v if len(v) else [0], dtype=np.uint64)
        keep = (a32(sched["rows"]), a32(sched["level_offsets"]), a64(sched["copy_src"]), a64(sched["copy_dst"]), a32(sched["copy_offsets"]))
        p32, p64 = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
        s = cls(int(sched["n_levels"]), keep[0].ctypes.data_as(p32), keep[1].ctypes.data_as(p32), keep[2].ctypes.data_as(p64),
                keep[3].ctypes.data_as(p64), keep[4].ctypes.data_as(p32))
        s._keep = keep
        return s


class Challenger(C.Structure):
    _fields_ = [("state", C.c_uint64 * 12), ("in_buf", C.c_uint64 * 8), ("n_in", C.c_uint64), ("out_buf", C.c_uint64 * 8),
                ("n_out", C.c_uint64)]


# every symbol include/sipp_hip.h declares: name -> (restype, argtypes)
u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)
vp = C.c_void_p
SIGNATURES = {
    "sipp_default_config": (None, [C.POINTER(StarkConfig)]),
    "sipp_ctx_create": (C.c_int, [C.POINTER(vp), C.c_int, C.POINTER(StarkConfig), C.c_size_t]),
    "sipp_ctx_destroy": (None, [vp]),
    "sipp_abi_version": (C.c_uint32, []),
    "sipp_ctx_create_checked": (C.c_int, [C.POINTER(vp), C.c_int, C.POINTER(StarkConfig), C.c_size_t, C.c_uint32, C.c_size_t]),
    "sipp_ctx_set_kernel_routes": (C.c_int, [vp, C.c_uint32]),
    "sipp_ctx_set_stream_priority": (C.c_int, [vp, C.c_int]),
    "sipp_last_error": (C.c_char_p, [vp]),
    "sipp_sync": (C.c_int, [vp]),
    "sipp_stream": (vp, [vp]),
    "sipp_g1_exp_prove": (C.c_int, [vp, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "sipp_g2_exp_prove": (C.c_int, [vp, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "sipp_fq12_exp_prove": (C.c_int, [vp, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "sipp_map_to_g2_prove": (C.c_int, [vp, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "sipp_pairing_prove": (C.c_int, [vp, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "sipp_prove": (C.c_int, [vp, C.c_int, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "sipp_ctx_set_hardened": (C.c_int, [vp, C.c_int]),
    "sipp_map_to_g2": (C.c_int, [vp, vp, C.c_size_t, vp, vp, vp]),
    "sipp_prove_async": (C.c_int, [vp, C.c_int, vp, C.c_size_t, vp, C.c_size_t]),
    "sipp_wait": (C.c_int, [vp, C.POINTER(C.c_size_t)]),
    "sipp_host_poseidon_permute": (C.c_int, [vp, C.c_size_t, C.c_int]),
    "sipp_instances_prove": (C.c_int, [C.POINTER(vp), C.c_size_t, C.c_size_t, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(vp),
                                       C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_int)]),
    "sipp_io_shard": (C.c_int, [C.c_size_t, C.c_uint32, C.c_uint32, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "sipp_instance_prove": (C.c_int, [C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(vp),
                                      C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "sipp_exp_outputs": (C.c_int, [vp, C.c_int, vp, C.c_size_t]),
    "sipp_inner_product": (C.c_int, [vp, vp, vp, C.c_size_t, vp]),
    "sipp_inner_products": (C.c_int, [vp, vp, vp, C.c_size_t, C.c_size_t, vp]),
    "sipp_native_proof_words": (C.c_size_t, [C.c_size_t]),
    "sipp_prove_native": (C.c_int, [vp, vp, vp, C.c_size_t, vp]),
    "sipp_verify_native": (C.c_int, [vp, vp, vp, C.c_size_t, vp, vp, vp, vp, vp, C.POINTER(C.c_int)]),
    "sipp_stark_verify": (C.c_int, [u64p, C.c_size_t, C.POINTER(StarkConfig), C.POINTER(C.c_int)]),
    "sipp_fri_verify_openings": (C.c_int, [u64p, C.c_size_t, C.POINTER(u64p), u32p, u32p, C.c_size_t, C.POINTER(FriBatch), C.c_size_t, C.c_uint32,
                                           C.POINTER(FriParams), C.POINTER(Challenger), C.POINTER(C.c_int)]),
    "sipp_plonk_verify_gates": (C.c_int, [u64p, C.c_size_t, u64p, C.POINTER(PlonkParams), C.POINTER(FriParams), C.POINTER(PlonkCircuit), u64p,
                                          C.POINTER(C.c_int)]),
    "sipp_proof_size": (C.c_size_t, [vp, C.c_int, C.c_size_t]),
    "sipp_stark_shape": (C.c_int, [vp, C.c_int, C.c_size_t, u32p, u32p, u32p, u32p]),
    "sipp_workspace_bytes": (C.c_size_t, [C.c_int, C.c_size_t]),
    "sipp_device_memory": (C.c_int, [C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "sipp_workspace_bytes_cfg": (C.c_size_t, [C.c_int, C.c_size_t, C.POINTER(StarkConfig)]),
    "sipp_fri_const_arity": (None, [C.POINTER(FriParams), C.c_uint32, C.c_uint32, C.c_uint32]),
    "sipp_commit_batch_ex": (C.c_int, [vp, vp, C.c_int, vp, vp, vp, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, vp, C.c_uint32, vp]),
    "sipp_fri_proof_size": (C.c_size_t, [C.POINTER(Oracle), C.c_size_t, C.POINTER(FriBatch), C.c_size_t, C.c_uint32, C.POINTER(FriParams)]),
    "sipp_fri_prove_openings": (C.c_int, [vp, C.POINTER(Oracle), C.c_size_t, C.POINTER(FriBatch), C.c_size_t, C.c_uint32,
                                          C.POINTER(FriParams), C.POINTER(Challenger), vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "sipp_plonk_num_partial_products": (C.c_uint32, [C.POINTER(PlonkParams)]),
    "sipp_plonk_zs_partial_products": (C.c_int, [vp, vp, vp, C.c_uint32, C.POINTER(PlonkParams), u64p, u64p, vp]),
    "sipp_plonk_quotient_chunks": (C.c_int, [vp, vp, vp, vp, C.c_uint32, C.c_uint32, C.POINTER(PlonkParams), u64p, u64p, u64p, vp]),
    "sipp_plonk_quotient_chunks_ex": (C.c_int, [vp, vp, vp, vp, C.c_uint32, C.c_uint32, C.POINTER(PlonkParams), u64p, u64p, u64p, vp, C.c_uint32, vp]),
    "sipp_plonk_prove_ex": (C.c_int, [vp, vp, vp, C.POINTER(Oracle), u64p, C.POINTER(Oracle), C.c_uint32, C.POINTER(PlonkParams), C.POINTER(FriParams),
                                      u64p, u64p, C.c_uint32, vp, C.c_uint32, vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "sipp_plonk_perm_proof_size": (C.c_size_t, [C.c_uint32, C.POINTER(PlonkParams), C.POINTER(FriParams)]),
    "sipp_plonk_gates_proof_size": (C.c_size_t, [C.c_uint32, C.POINTER(PlonkParams), C.POINTER(FriParams), C.POINTER(PlonkCircuit), C.c_uint32]),
    "sipp_plonk_prove_gates": (C.c_int, [vp, vp, vp, C.POINTER(Oracle), u64p, C.POINTER(Oracle), C.c_uint32, C.POINTER(PlonkParams), C.POINTER(FriParams),
                                         C.POINTER(PlonkCircuit), u64p, u64p, C.c_uint32, vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "sipp_plonk_generate_witness": (C.c_int, [vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(PlonkGenerator), C.c_size_t, u64p]),
    "sipp_plonk_generate_witness_levels": (C.c_int, [vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(PlonkGenerator), C.c_size_t, u64p,
                                                     C.POINTER(PlonkSchedule)]),
    "sipp_circuit_workspace_bytes": (C.c_size_t, [C.c_uint32, C.POINTER(PlonkParams), C.POINTER(FriParams), C.POINTER(PlonkCircuit)]),
    "sipp_circuit_build": (C.c_int, [vp, C.c_uint32, C.POINTER(PlonkParams), C.POINTER(FriParams), C.POINTER(PlonkCircuit), vp, C.POINTER(PlonkGenerator),
                                     C.c_size_t, C.POINTER(PlonkScheduleHost), u64p, C.POINTER(vp)]),
    "sipp_circuit_destroy": (None, [vp]),
    "sipp_circuit_verifier_data": (C.c_int, [vp, vp, vp]),
    "sipp_circuit_proof_size": (C.c_size_t, [vp, C.c_uint32]),
    "sipp_circuit_prove": (C.c_int, [vp, vp, u64p, C.c_uint32, vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "sipp_circuit_verify": (C.c_int, [vp, vp, C.c_size_t, C.POINTER(C.c_int)]),
    "sipp_plonk_perm_prove": (C.c_int, [vp, vp, vp, C.c_uint32, C.POINTER(PlonkParams), C.POINTER(FriParams), u64p, u64p, vp, C.c_size_t,
                                        C.POINTER(C.c_size_t)]),
    "sipp_ntt_batch": (C.c_int, [vp, vp, C.c_size_t, C.c_size_t, C.c_uint32, C.c_int]),
    "sipp_lde_batch": (C.c_int, [vp, vp, vp, vp, C.c_size_t, C.c_uint32]),
    "sipp_poseidon_leaves": (C.c_int, [vp, vp, C.c_size_t, C.c_uint32, vp]),
    "sipp_merkle_cap": (C.c_int, [vp, vp, C.c_uint32, vp]),
    "sipp_commit_batch": (C.c_int, [vp, vp, vp, vp, vp, C.c_size_t, C.c_uint32, vp]),
    "sipp_poseidon_permute": (C.c_int, [vp, vp, C.c_size_t]),
    "sipp_trace_build": (C.c_int, [vp, C.c_int, vp, C.c_size_t, vp]),
    "sipp_profile_enable": (C.c_int, [vp, C.c_int]),
    "sipp_profile_reset": (C.c_int, [vp]),
    "sipp_profile_report": (C.c_int, [vp, C.c_char_p, C.c_size_t]),
    "sipp_timer_start": (C.c_int, [vp]),
    "sipp_timer_stop": (C.c_int, [vp, C.POINTER(C.c_float)]),
}

_lib = None


def lib():
    """Load libsipp_hip.so.  No fallback: a missing library is an error."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO):
        raise SippError(-7, "libsipp_hip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    # torch ships its own libamdhip64: load torch FIRST so that libsipp_hip.so binds to the same HIP runtime
    # (two runtimes in one process make hipGetDeviceCount fail in whichever comes second)
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(SO)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


SIPP_E_VERIFY = -9


def stark_verify(proof, cfg=None):
    """sipp_stark_verify: the library's own check of a flat proof (host code; no ctx, no GPU).  Returns 0 for an accepted proof, else the
    stage that refused it (include/sipp_hip.h lists them: 110 = the constraints do not meet the quotient, 123 = a Merkle path ...).
    Raises SippError for a call the library cannot serve (SIPP_E_BADARG / SIPP_E_UNSUPPORTED)."""
    pf = np.ascontiguousarray(proof, dtype=np.uint64)
    reason = C.c_int(0)
    rc = lib().sipp_stark_verify(pf.ctypes.data_as(u64p), len(pf), C.byref(cfg) if cfg is not None else None, C.byref(reason))
    if rc not in (0, SIPP_E_VERIFY):
        raise SippError(rc, "sipp_stark_verify")
    return reason.value


def default_config():
    cfg = StarkConfig()
    lib().sipp_default_config(C.byref(cfg))
    return cfg


def device_memory_bytes(device=0):
    """total memory of GPU `device` (sipp_device_memory)"""
    total = C.c_size_t()
    rc = lib().sipp_device_memory(int(device), None, C.byref(total))
    if rc != 0:
        raise SippError(rc, "device_memory")
    return total.value


def device_free_bytes(device=0):
    """free memory of GPU `device` right now (sipp_device_memory)"""
    free = C.c_size_t()
    rc = lib().sipp_device_memory(int(device), C.byref(free), None)
    if rc != 0:
        raise SippError(rc, "device_memory")
    return free.value


def to_device(arr):
    """host uint64 ndarray -> device tensor (int64 bit pattern)"""
    import torch
    a = np.ascontiguousarray(arr, dtype=np.uint64)
    return torch.from_numpy(a.view(np.int64)).to("cuda")


def to_host(t):
    return t.detach().cpu().numpy().view(np.uint64)


class Ctx:
    """One GPU, one stream, one workspace arena (sipp_ctx)."""

    def __init__(self, device=0, cfg=None, workspace_bytes=1 << 30):
        import torch
        if not torch.cuda.is_available():
            raise SippError(-2, "no GPU visible: the SIPP HIP path has no CPU fallback")
        self.L = lib()
        self.device = device
        torch.cuda.set_device(device)
        h = vp()
        rc = self.L.sipp_ctx_create(C.byref(h), device, C.byref(cfg) if cfg is not None else None, workspace_bytes)
        if rc != 0:
            raise SippError(rc, "sipp_ctx_create")
        self.h = h
        self.workspace_bytes = int(workspace_bytes) if workspace_bytes else 24 << 30      # the arena this ctx reserved (0 = the library's default)
        self.cfg = cfg if cfg is not None else default_config()

    def close(self):
        if getattr(self, "h", None):
            self.L.sipp_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        if rc != 0:
            raise SippError(rc, "%s: %s" % (what, self.L.sipp_last_error(self.h).decode()))

    def sync(self):
        self._ck(self.L.sipp_sync(self.h), "sync")

    # ---- building blocks (device tensors) ----
    def ntt(self, t, log_n, inverse=False):
        """in place, natural -> natural; t is [ncols, stride] int64"""
        assert t.is_cuda and t.dim() == 2 and t.is_contiguous()
        self._ck(self.L.sipp_ntt_batch(self.h, t.data_ptr(), t.shape[1], t.shape[0], log_n, int(inverse)), "ntt_batch")
        return t

    def lde(self, values, log_n):
        import torch
        ncols = values.shape[0]
        rb = self.cfg.rate_bits
        coeffs = torch.empty_like(values)
        lde = torch.empty((ncols, values.shape[1] << rb), dtype=torch.int64, device=values.device)
        self._ck(self.L.sipp_lde_batch(self.h, values.data_ptr(), coeffs.data_ptr(), lde.data_ptr(), ncols, log_n),
                 "lde_batch")
        return coeffs, lde

    def poseidon_leaves(self, lde, log_leaves):
        import torch
        dig = torch.empty(((1 << log_leaves), 4), dtype=torch.int64, device=lde.device)
        self._ck(self.L.sipp_poseidon_leaves(self.h, lde.data_ptr(), lde.shape[0], log_leaves, dig.data_ptr()),
                 "poseidon_leaves")
        return dig

    def merkle_cap(self, tree, log_leaves):
        ch = min(self.cfg.cap_height, log_leaves)
        cap = np.zeros((1 << ch, 4), dtype=np.uint64)
        self._ck(self.L.sipp_merkle_cap(self.h, tree.data_ptr(), log_leaves, cap.ctypes.data), "merkle_cap")
        return cap

    def commit(self, values, log_n, bufs=None):
        """PolynomialBatch::from_values on device.  Returns (coeffs, lde, tree, cap)."""
        import torch
        ncols = values.shape[0]
        rb = self.cfg.rate_bits
        log_m = log_n + rb
        if bufs is None:
            coeffs = torch.empty_like(values)
            lde = torch.empty((ncols, 1 << log_m), dtype=torch.int64, device=values.device)
            tree = torch.empty((2 << log_m, 4), dtype=torch.int64, device=values.device)
        else:
            coeffs, lde, tree = bufs
        ch = min(self.cfg.cap_height, log_m)
        cap = np.zeros((1 << ch, 4), dtype=np.uint64)
        self._ck(self.L.sipp_commit_batch(self.h, values.data_ptr(), coeffs.data_ptr(), lde.data_ptr(),
                                          tree.data_ptr(), ncols, log_n, cap.ctypes.data), "commit_batch")
        return coeffs, lde, tree, cap

    # ---- generic PolynomialBatch / FRI opening proofs ----
    def commit_ex(self, data, log_n, rate_bits, cap_height, from_coeffs=False, salt=None):
        """sipp_commit_batch_ex: data [ncols, N] device int64 (values or coefficients), salt [4, N << rate_bits] or None.
        Returns (Oracle struct, cap ndarray, (coeffs, lde, tree) tensors kept alive by the caller)."""
        import torch
        ncols = data.shape[0]
        n_salt = 0 if salt is None else salt.shape[0]
        m = 1 << (log_n + rate_bits)
        coeffs = torch.empty_like(data)
        lde = torch.empty((ncols + n_salt, m), dtype=torch.int64, device=data.device)
        tree = torch.empty((2 * m, 4), dtype=torch.int64, device=data.device)
        ch = min(cap_height, log_n + rate_bits)
        cap = np.zeros((1 << ch, 4), dtype=np.uint64)
        self._ck(self.L.sipp_commit_batch_ex(self.h, data.data_ptr(), int(from_coeffs), coeffs.data_ptr(), lde.data_ptr(), tree.data_ptr(),
                                             ncols, log_n, rate_bits, cap_height, None if salt is None else salt.data_ptr(), n_salt,
                                             cap.ctypes.data), "commit_batch_ex")
        return Oracle(coeffs.data_ptr(), lde.data_ptr(), tree.data_ptr(), ncols, n_salt), cap, (coeffs, lde, tree)

    def fri_prove_openings(self, oracles, batches, log_n, params, challenger):
        """sipp_fri_prove_openings.  oracles: [Oracle], batches: [((c0, c1), [(oracle, col_begin, col_end), ...])];
        `challenger` (Challenger struct) is advanced in place.  Returns the flat proof."""
        oa = (Oracle * len(oracles))(*oracles)
        ba = (FriBatch * len(batches))()
        keep = []
        for i, (pt, ranges) in enumerate(batches):
            r = (PolyRange * len(ranges))(*[PolyRange(*x) for x in ranges])
            keep.append(r)
            ba[i].point[0], ba[i].point[1] = int(pt[0]), int(pt[1])
            ba[i].n_ranges = len(ranges)
            ba[i].ranges = r
        cap = self.L.sipp_fri_proof_size(oa, len(oracles), ba, len(batches), log_n, C.byref(params))
        if cap == 0:
            raise SippError(-1, "sipp_fri_proof_size")
        out = np.zeros(cap, dtype=np.uint64)
        n = C.c_size_t()
        self._ck(self.L.sipp_fri_prove_openings(self.h, oa, len(oracles), ba, len(batches), log_n, C.byref(params),
                                                C.byref(challenger), out.ctypes.data, cap, C.byref(n)), "fri_prove_openings")
        return out[: n.value]

    # ---- plonky2's wire permutation argument ----
    @staticmethod
    def _u64(v):
        a = (C.c_uint64 * len(v))(*[int(x) for x in v])
        return a

    def plonk_zs(self, wires, sigmas, log_n, p, betas, gammas):
        """sipp_plonk_zs_partial_products: wires, sigmas [R, N] device int64 -> [C (1 + num_prods), N]"""
        import torch
        rows = p.num_challenges * (1 + self.L.sipp_plonk_num_partial_products(C.byref(p)))
        out = torch.empty((rows, 1 << log_n), dtype=torch.int64, device=wires.device)
        self._ck(self.L.sipp_plonk_zs_partial_products(self.h, wires.data_ptr(), sigmas.data_ptr(), log_n, C.byref(p), self._u64(betas),
                                                       self._u64(gammas), out.data_ptr()), "plonk_zs_partial_products")
        return out

    def plonk_quotient_chunks(self, wires_lde, sigmas_lde, zs_lde, log_n, rate_bits, p, betas, gammas, alphas):
        import torch
        out = torch.empty((p.num_challenges * p.max_degree, 1 << log_n), dtype=torch.int64, device=wires_lde.device)
        self._ck(self.L.sipp_plonk_quotient_chunks(self.h, wires_lde.data_ptr(), sigmas_lde.data_ptr(), zs_lde.data_ptr(), log_n, rate_bits,
                                                   C.byref(p), self._u64(betas), self._u64(gammas), self._u64(alphas), out.data_ptr()),
                 "plonk_quotient_chunks")
        return out

    def plonk_quotient_chunks_ex(self, wires_lde, sigmas_lde, zs_lde, log_n, rate_bits, p, betas, gammas, alphas, gate_terms):
        """gate_terms: [K, N << rate_bits] device int64 (leaf order) or None"""
        import torch
        out = torch.empty((p.num_challenges * p.max_degree, 1 << log_n), dtype=torch.int64, device=wires_lde.device)
        k = 0 if gate_terms is None else gate_terms.shape[0]
        self._ck(self.L.sipp_plonk_quotient_chunks_ex(self.h, wires_lde.data_ptr(), sigmas_lde.data_ptr(), zs_lde.data_ptr(), log_n, rate_bits,
                                                      C.byref(p), self._u64(betas), self._u64(gammas), self._u64(alphas),
                                                      None if gate_terms is None else gate_terms.data_ptr(), k, out.data_ptr()),
                 "plonk_quotient_chunks_ex")
        return out

    def plonk_prove_ex(self, wires, sigmas, log_n, p, fp, digest, public_inputs, gate_terms=None, wires_oracle=None, wires_cap=None,
                       sigmas_oracle=None):
        """sipp_plonk_prove_ex: the flow with gate terms, public inputs (host ints) and optional pre-committed oracles (Oracle structs)"""
        pis = [int(x) for x in public_inputs]
        cap = self.L.sipp_plonk_perm_proof_size(log_n, C.byref(p), C.byref(fp))
        if cap == 0:
            raise SippError(-1, "sipp_plonk_perm_proof_size")
        cap += len(pis)
        out = np.zeros(cap, dtype=np.uint64)
        n = C.c_size_t()
        k = 0 if gate_terms is None else gate_terms.shape[0]
        wc_arr = None if wires_cap is None else np.ascontiguousarray(wires_cap, dtype=np.uint64).reshape(-1)     # alive until the call returns
        wcap = None if wc_arr is None else wc_arr.ctypes.data_as(u64p)
        self._ck(self.L.sipp_plonk_prove_ex(self.h, wires.data_ptr(), sigmas.data_ptr(), None if wires_oracle is None else C.byref(wires_oracle), wcap,
                                            None if sigmas_oracle is None else C.byref(sigmas_oracle), log_n, C.byref(p), C.byref(fp),
                                            self._u64(digest), self._u64(pis) if pis else None, len(pis),
                                            None if gate_terms is None else gate_terms.data_ptr(), k, out.ctypes.data, cap, C.byref(n)),
                 "plonk_prove_ex")
        return out[: n.value]

    def plonk_prove_gates(self, wires, constants_sigmas, log_n, p, fp, circuit, digest, public_inputs, wires_oracle=None, wires_cap=None,
                          cs_oracle=None):
        """sipp_plonk_prove_gates: wires [num_wires][N], constants_sigmas [num_constants + num_routed][N] device tensors (values);
        circuit = PlonkCircuit; the gate constraints are interpreted inside the quotient kernel.  Returns the flat "SIPPPLK3" proof."""
        pis = [int(x) for x in public_inputs]
        cap = self.L.sipp_plonk_gates_proof_size(log_n, C.byref(p), C.byref(fp), C.byref(circuit), len(pis))
        if cap == 0:
            raise SippError(-1, "sipp_plonk_gates_proof_size")
        out = np.zeros(cap, dtype=np.uint64)
        n = C.c_size_t()
        wc_arr = None if wires_cap is None else np.ascontiguousarray(wires_cap, dtype=np.uint64).reshape(-1)
        wcap = None if wc_arr is None else wc_arr.ctypes.data_as(u64p)
        self._ck(self.L.sipp_plonk_prove_gates(self.h, wires.data_ptr(), constants_sigmas.data_ptr(),
                                               None if wires_oracle is None else C.byref(wires_oracle), wcap,
                                               None if cs_oracle is None else C.byref(cs_oracle), log_n, C.byref(p), C.byref(fp), C.byref(circuit),
                                               self._u64(digest), self._u64(pis) if pis else None, len(pis), out.ctypes.data, cap, C.byref(n)),
                 "plonk_prove_gates")
        return out[: n.value]

    def plonk_generate_witness(self, wires, constants, log_n, gens, pih=None):
        """sipp_plonk_generate_witness: the gates' witness generators, in place on the device wire table [num_wires][N]; constants
        [>= num_constants][N] device tensor (the front of constants_sigmas); gens = [(kind, selector_index, row, p0 .. p4)]"""
        arr = (PlonkGenerator * len(gens))(*[PlonkGenerator(int(g[0]), int(g[1]), int(g[2]), (C.c_uint32 * 5)(*[int(x) for x in g[3:8]])) for g in gens])
        self._ck(self.L.sipp_plonk_generate_witness(self.h, wires.data_ptr(), constants.data_ptr(), log_n, wires.shape[0], constants.shape[0], arr,
                                                    len(gens), None if pih is None else self._u64([int(x) for x in pih])), "plonk_generate_witness")

    def plonk_generate_witness_levels(self, wires, constants, log_n, gens, pih, sched):
        """sipp_plonk_generate_witness_levels: the generators level by level with the copies between levels (sched = PlonkSchedule);
        synchronises the ctx stream (the schedule's bounds are checked on the device)"""
        arr = (PlonkGenerator * len(gens))(*[PlonkGenerator(int(g[0]), int(g[1]), int(g[2]), (C.c_uint32 * 5)(*[int(x) for x in g[3:8]])) for g in gens])
        self._ck(self.L.sipp_plonk_generate_witness_levels(self.h, wires.data_ptr(), constants.data_ptr(), log_n, wires.shape[0], constants.shape[0],
                                                           arr, len(gens), None if pih is None else self._u64([int(x) for x in pih]), C.byref(sched)),
                 "plonk_generate_witness_levels")

    def plonk_perm_prove(self, wires, sigmas, log_n, p, fp, digest=(1, 2, 3, 4), pih=(0, 0, 0, 0)):
        cap = self.L.sipp_plonk_perm_proof_size(log_n, C.byref(p), C.byref(fp))
        if cap == 0:
            raise SippError(-1, "sipp_plonk_perm_proof_size")
        out = np.zeros(cap, dtype=np.uint64)
        n = C.c_size_t()
        self._ck(self.L.sipp_plonk_perm_prove(self.h, wires.data_ptr(), sigmas.data_ptr(), log_n, C.byref(p), C.byref(fp), self._u64(digest),
                                              self._u64(pih), out.ctypes.data, cap, C.byref(n)), "plonk_perm_prove")
        return out[: n.value]

    def shape(self, kind, num_io):
        a, b, c, d = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._ck(self.L.sipp_stark_shape(self.h, kind, num_io, C.byref(a), C.byref(b), C.byref(c), C.byref(d)), "stark_shape")
        return a.value, b.value, c.value, d.value

    def trace_build(self, kind, ios):
        """device trace [W][N] (natural row order) for the IO records (host uint32 [num_io][words])"""
        import torch
        ios = np.ascontiguousarray(ios, dtype=np.uint32)
        log_n, W, _, _ = self.shape(kind, ios.shape[0])
        t = torch.empty((W, 1 << log_n), dtype=torch.int64, device="cuda")
        self._ck(self.L.sipp_trace_build(self.h, kind, ios.ctypes.data, ios.shape[0], t.data_ptr()), "trace_build")
        return t

    def prove(self, kind, ios):
        """one STARK sub-proof (kind 0 G1 / 1 G2 / 2 Fq12 / 3 MapToG2 / 4, 5 hardened G1, G2 / 6 final pairing) for host IO records -> flat proof"""
        ios = np.ascontiguousarray(ios, dtype=np.uint32)
        cap = self.L.sipp_proof_size(self.h, kind, ios.shape[0])
        if cap == 0:
            raise SippError(-1, "sipp_proof_size")
        out = np.zeros(cap, dtype=np.uint64)
        n = C.c_size_t()
        if kind == 6:
            self._ck(self.L.sipp_pairing_prove(self.h, ios.ctypes.data, ios.shape[0], out.ctypes.data, cap, C.byref(n)), "pairing_prove")
            return out[: n.value]
        if kind >= 4:       # the hardened G1 / G2 kinds: the generic entry point
            self._ck(self.L.sipp_prove(self.h, kind, ios.ctypes.data, ios.shape[0], out.ctypes.data, cap, C.byref(n)), "prove")
            return out[: n.value]
        fn = (self.L.sipp_g1_exp_prove, self.L.sipp_g2_exp_prove, self.L.sipp_fq12_exp_prove, self.L.sipp_map_to_g2_prove)[kind]
        self._ck(fn(self.h, ios.ctypes.data, ios.shape[0], out.ctypes.data, cap, C.byref(n)), "prove")
        return out[: n.value]

    def exp_outputs(self, kind, ios):
        """copy of the records with the output words computed on the device (sipp_exp_outputs)"""
        rec = np.array(ios, dtype=np.uint32, order="C", copy=True)
        self._ck(self.L.sipp_exp_outputs(self.h, kind, rec.ctypes.data, rec.shape[0]), "exp_outputs")
        return rec

    def map_to_g2(self, msgs, cofactor=True):
        """messages [n][16] (u in Fp2) -> (MapToG2 records [n][48], G2ExpStark records of the cofactor clearing [2n][104],
        cleared points [n][32]) -- sipp_map_to_g2; without `cofactor` only the records"""
        msgs = np.ascontiguousarray(msgs, dtype=np.uint32).reshape(-1, 16)
        n = msgs.shape[0]
        recs = np.zeros((n, 48), dtype=np.uint32)
        if not cofactor:
            self._ck(self.L.sipp_map_to_g2(self.h, msgs.ctypes.data, n, recs.ctypes.data, None, None), "map_to_g2")
            return recs
        g2 = np.zeros((2 * n, 104), dtype=np.uint32)
        pts = np.zeros((n, 32), dtype=np.uint32)
        self._ck(self.L.sipp_map_to_g2(self.h, msgs.ctypes.data, n, recs.ctypes.data, g2.ctypes.data, pts.ctypes.data), "map_to_g2")
        return recs, g2, pts

    def inner_products(self, g1, g2, count=1):
        """prod_i pairing(A_i, B_i) (sipp_inner_products): g1 [count * n, 16], g2 [count * n, 32] uint32 limbs -> [count, 96]"""
        g1 = np.ascontiguousarray(g1, dtype=np.uint32).reshape(-1, 16)
        g2 = np.ascontiguousarray(g2, dtype=np.uint32).reshape(-1, 32)
        assert g1.shape[0] == g2.shape[0] and g1.shape[0] % count == 0
        out = np.zeros((count, 96), dtype=np.uint32)
        self._ck(self.L.sipp_inner_products(self.h, g1.ctypes.data, g2.ctypes.data, g1.shape[0] // count, count,
                                            out.ctypes.data), "inner_products")
        return out

    def prove_native(self, A, B):
        """sipp_prove_native: A [n, 16], B [n, 32] uint32 limbs -> proof [2 log2 n + 1, 96]"""
        A = np.ascontiguousarray(A, dtype=np.uint32).reshape(-1, 16)
        B = np.ascontiguousarray(B, dtype=np.uint32).reshape(-1, 32)
        n = A.shape[0]
        words = self.L.sipp_native_proof_words(n)
        if words == 0 or B.shape[0] != n:
            raise SippError(-1, "prove_native: n must be a power of two")
        proof = np.zeros((words // 96, 96), dtype=np.uint32)
        self._ck(self.L.sipp_prove_native(self.h, A.ctypes.data, B.ctypes.data, n, proof.ctypes.data), "prove_native")
        return proof

    def verify_native(self, A, B, proof):
        """sipp_verify_native -> (accepted, statement, [g1_ios, g2_ios, fq12_ios])"""
        A = np.ascontiguousarray(A, dtype=np.uint32).reshape(-1, 16)
        B = np.ascontiguousarray(B, dtype=np.uint32).reshape(-1, 32)
        proof = np.ascontiguousarray(proof, dtype=np.uint32)
        n = A.shape[0]
        lg = n.bit_length() - 1
        st = np.zeros(48 * n + 240, dtype=np.uint32)
        ios = [np.zeros((max(n - 1, 0), 56), dtype=np.uint32), np.zeros((max(n - 1, 0), 104), dtype=np.uint32),
               np.zeros((2 * lg, 296), dtype=np.uint32)]
        ok = C.c_int(0)
        self._ck(self.L.sipp_verify_native(self.h, A.ctypes.data, B.ctypes.data, n, proof.ctypes.data, st.ctypes.data,
                                           ios[0].ctypes.data, ios[1].ctypes.data, ios[2].ctypes.data, C.byref(ok)),
                 "verify_native")
        return bool(ok.value), st, ios

    def prove_async(self, kind, ios):
        """start one sub-proof on the ctx's worker thread (sipp_prove_async); collect it with wait()"""
        ios = np.ascontiguousarray(ios, dtype=np.uint32)
        cap = self.L.sipp_proof_size(self.h, kind, ios.shape[0])
        if cap == 0:
            raise SippError(-1, "sipp_proof_size")
        out = np.zeros(cap, dtype=np.uint64)
        self._ck(self.L.sipp_prove_async(self.h, kind, ios.ctypes.data, ios.shape[0], out.ctypes.data, cap), "prove_async")
        self._inflight = (ios, out)      # both buffers stay alive until wait()
        return self

    def wait(self):
        n = C.c_size_t()
        rc = self.L.sipp_wait(self.h, C.byref(n))
        ios, out = getattr(self, "_inflight", None) or (None, None)
        self._inflight = None
        self._ck(rc, "wait")
        return out[: n.value]

    def poseidon_permute(self, states):
        self._ck(self.L.sipp_poseidon_permute(self.h, states.data_ptr(), states.shape[0]), "poseidon_permute")
        return states

    # ---- measurement ----
    def profile(self, enable=True):
        self._ck(self.L.sipp_profile_enable(self.h, int(enable)), "profile_enable")

    def profile_reset(self):
        self._ck(self.L.sipp_profile_reset(self.h), "profile_reset")

    def profile_report(self):
        buf = C.create_string_buffer(1 << 16)
        self._ck(self.L.sipp_profile_report(self.h, buf, len(buf)), "profile_report")
        return json.loads(buf.value.decode())

    def timer_start(self):
        self._ck(self.L.sipp_timer_start(self.h), "timer_start")

    def timer_stop(self):
        ms = C.c_float()
        self._ck(self.L.sipp_timer_stop(self.h, C.byref(ms)), "timer_stop")
        return ms.value


class Instance:
    """The three sub-proofs of one SIPP instance (reference src/verifier_circuit.rs:133-135) on three ctxs = three HIP
    streams, started together through sipp_instance_prove.  `devices` may name one GPU (default) or three
    (level L-B).  `priorities`: stream level per kind ("high" = a high-priority stream; "low" / "" = a stream with a hardware queue of its own at
    normal priority, see sipp_ctx_set_stream_priority in include/sipp_hip.h); G1 low / G2 normal / Fq12 high
    measured best with the start gate of sipp_instance_prove (69.3 vs 71.2 ms per n = 128 instance for low / high / high)."""

    # three concurrent arenas above this share of the card's memory -> one ctx, the proofs back to back (sipp_instance_prove with
    # three equal handles): of the BASELINE configs only the HARDENED n = 4096 instance on ONE GPU (276 GB of 288 GB; the plain one's
    # 246 GB fit).  Measured (scripts/large_n_modes.py, round 4): back to back costs 5.5 % at n = 4096 (1234 against 1169 ms, plain) and
    # 13 - 15 % at n = 1024 (357 against 311 ms) -- so three streams wherever they fit.
    SINGLE_CTX_SHARE = 0.9

    def __init__(self, num_io, devices=(0, 0, 0), priorities=None, hardened=False, single_ctx=None):
        """hardened: G1 / G2 with the hardened AIRs (kinds 4 / 5 in the proofs' headers; sipp_ctx_set_hardened).
        single_ctx: True = one ctx / one arena, the three proofs back to back; False = three ctxs on three streams; None = three
        unless their arenas together exceed SINGLE_CTX_SHARE of the device's memory (and the devices are one device).
        priorities=None: ("low", "", "high") for both AIR variants.  (Until the thin trees' hasher moved to the matrix pipe in round 4 a
        hardened instance wanted G2 on a high-priority stream too: 64.0 - 64.2 against 66.9 - 68.1 ms.  With the shorter Fq12 chain the
        settings are level -- scripts/sweep_prios.sh, two passes: single 56.4 - 57.5 for both, five queued 49.7 - 49.9 against 50.9 ms.)"""
        if priorities is None:
            priorities = ("low", "", "high")
        self.L = lib()
        self.num_io = tuple(int(x) for x in num_io)
        self.ctxs = []
        level = {"low": -1, "": 0, "normal": 0, "high": 1}
        ws = [self.L.sipp_workspace_bytes(k + 4 if (hardened and k < 2) else k, max(1, self.num_io[k])) for k in range(3)]
        if single_ctx is None:
            # three arenas only where they fit what is FREE on the card now (other instances, a queue's earlier slots, torch's pool)
            single_ctx = len(set(devices)) == 1 and sum(ws) > min(self.SINGLE_CTX_SHARE * device_memory_bytes(devices[0]),
                                                                  device_free_bytes(devices[0]))
        self.single_ctx = bool(single_ctx)
        try:
            if self.single_ctx:
                c = Ctx(device=devices[0], workspace_bytes=max(ws))
                self.ctxs = [c, c, c]
                if hardened:
                    c._ck(self.L.sipp_ctx_set_hardened(c.h, 1), "set_hardened")
            else:
                for k in range(3):
                    c = Ctx(device=devices[k], workspace_bytes=ws[k])
                    self.ctxs.append(c)
                    if hardened:
                        c._ck(self.L.sipp_ctx_set_hardened(c.h, 1), "set_hardened")
                    c._ck(self.L.sipp_ctx_set_stream_priority(c.h, level[priorities[k]]), "set_stream_priority")
        except Exception:
            self.close()          # the ctxs (and their arenas) created so far
            raise
        self.caps = [self.L.sipp_proof_size(self.ctxs[k].h, k, self.num_io[k]) if self.num_io[k] else 0 for k in range(3)]
        self.out = [np.zeros(max(c, 1), dtype=np.uint64) for c in self.caps]

    def prove(self, ios):
        """ios: [g1, g2, fq12] host uint32 arrays -> three flat proofs (views of this object's output buffers); a kind with
        no record (an IO shard may have none) yields an empty proof"""
        ios = [np.ascontiguousarray(a, dtype=np.uint32) for a in ios]
        assert tuple(a.shape[0] for a in ios) == self.num_io
        h = (vp * 3)(*[c.h for c in self.ctxs])
        pi = (vp * 3)(*[a.ctypes.data for a in ios])
        ni = (C.c_size_t * 3)(*self.num_io)
        po = (vp * 3)(*[o.ctypes.data for o in self.out])
        pc = (C.c_size_t * 3)(*self.caps)
        pl = (C.c_size_t * 3)()
        rc = self.L.sipp_instance_prove(h, pi, ni, po, pc, pl)
        if rc != 0:
            msgs = "; ".join(self.L.sipp_last_error(c.h).decode() for c in self.ctxs)
            raise SippError(rc, "instance_prove: " + msgs)
        return [self.out[k][: pl[k]] for k in range(3)]

    def distinct_ctxs(self):
        return self.ctxs[:1] if getattr(self, "single_ctx", False) else list(self.ctxs)

    def sync(self):
        for c in self.distinct_ctxs():
            c.sync()

    def close(self):
        for c in self.distinct_ctxs():
            c.close()
        self.ctxs = []


class InstanceQueue:
    """sipp_instances_prove: `in_flight` slots of three ctxs on one device; prove(list of [g1, g2, fq12]) proves every instance of
    the list, `in_flight` at a time, and returns their proofs (copies)."""

    def __init__(self, num_io, in_flight=3, device=0, priorities=None, hardened=False):
        """priorities=None: Instance's own default"""
        self.L = lib()
        self.num_io = tuple(int(x) for x in num_io)
        self.slots = []
        # every slot needs three DISTINCT ctxs (sipp_instances_prove refuses duplicate handles): no one-ctx fallback here -- say so
        # before the first hipMalloc instead of failing in the middle of the slots
        need = in_flight * sum(self.L.sipp_workspace_bytes(k + 4 if (hardened and k < 2) else k, max(1, self.num_io[k])) for k in range(3))
        free = device_free_bytes(device)
        if need > free:
            raise SippError(-3, "InstanceQueue: %d slots x three arenas need %.1f GB, %.1f GB are free on device %d; lower in_flight "
                                "(or prove such instances one at a time through Instance, which falls back to one arena)"
                            % (in_flight, need / 2**30, free / 2**30, device))
        try:
            for _ in range(in_flight):
                self.slots.append(Instance(self.num_io, devices=(device,) * 3, priorities=priorities, hardened=hardened, single_ctx=False))
        except Exception:
            self.close()
            raise

    def prove(self, instances):
        count, F = len(instances), len(self.slots)
        keep = [[np.ascontiguousarray(a, dtype=np.uint32) for a in inst] for inst in instances]
        for inst in keep:
            assert tuple(a.shape[0] for a in inst) == self.num_io
        caps = self.slots[0].caps
        outs = [[np.zeros(max(c, 1), dtype=np.uint64) for c in caps] for _ in range(count)]
        h = (vp * (3 * F))(*[c.h for s in self.slots for c in s.ctxs])
        pi = (vp * (3 * count))(*[a.ctypes.data for inst in keep for a in inst])
        ni = (C.c_size_t * (3 * count))(*[n for _ in range(count) for n in self.num_io])
        po = (vp * (3 * count))(*[o.ctypes.data for inst in outs for o in inst])
        pc = (C.c_size_t * (3 * count))(*[c for _ in range(count) for c in caps])
        pl = (C.c_size_t * (3 * count))()
        st = (C.c_int * count)()
        rc = self.L.sipp_instances_prove(h, F, count, pi, ni, po, pc, pl, st)
        if rc != 0:
            msgs = "; ".join(self.L.sipp_last_error(c.h).decode() for s in self.slots for c in s.ctxs)
            raise SippError(rc, "instances_prove: " + msgs)
        return [[outs[i][k][: pl[3 * i + k]] for k in range(3)] for i in range(count)]

    def close(self):
        for s in self.slots:
            s.close()
        self.slots = []



class CircuitData:
    """plonky2's CircuitData for the outer proof over HOST arrays (sipp_circuit_build / _prove / _verify: reference
    src/verifier_circuit.rs:225, :253, :254): the gate set, the generators with their level schedule and the constants_sigmas values go in
    once; prove() takes the wire table with its input cells set and the public inputs, both numpy, and returns the flat proof."""

    def __init__(self, ctx, log_n, params, fri, circuit, constants_sigmas, gens, sched=None, digest=None):
        self.ctx, self.L = ctx, ctx.L                      # the ctx must outlive this object
        self.circuit, self.params, self.fri = circuit, params, fri
        cs = np.ascontiguousarray(constants_sigmas, dtype=np.uint64)
        arr = (PlonkGenerator * max(1, len(gens)))(*[PlonkGenerator(int(g[0]), int(g[1]), int(g[2]), (C.c_uint32 * 5)(*[int(x) for x in g[3:8]])) for g in gens])
        self._sched = None if sched is None else PlonkScheduleHost.from_dict(sched)
        h = C.c_void_p()
        rc = self.L.sipp_circuit_build(ctx.h, log_n, C.byref(params), C.byref(fri), C.byref(circuit), cs.ctypes.data, arr, len(gens),
                                       None if self._sched is None else C.byref(self._sched),
                                       None if digest is None else ctx._u64([int(x) for x in digest]), C.byref(h))
        ctx._ck(rc, "circuit_build")
        self.h, self.num_wires, self.n = h, circuit.num_wires, 1 << log_n
        cap = np.zeros(((1 << min(fri.cap_height, log_n + fri.rate_bits)), 4), dtype=np.uint64)
        dg = np.zeros(4, dtype=np.uint64)
        ctx._ck(self.L.sipp_circuit_verifier_data(self.h, cap.ctypes.data, dg.ctypes.data), "circuit_verifier_data")
        self.cap, self.digest = cap, dg

    def prove(self, wires, public_inputs):
        w = np.ascontiguousarray(wires, dtype=np.uint64)
        if w.shape != (self.num_wires, self.n):
            raise SippError(-1, "CircuitData.prove: wires must be [num_wires][N]")
        pis = [int(x) for x in public_inputs]
        cap = self.L.sipp_circuit_proof_size(self.h, len(pis))
        if cap == 0:
            raise SippError(-1, "sipp_circuit_proof_size")
        out = np.zeros(cap, dtype=np.uint64)
        n = C.c_size_t()
        self.ctx._ck(self.L.sipp_circuit_prove(self.h, w.ctypes.data, self.ctx._u64(pis) if pis else None, len(pis), out.ctypes.data, cap, C.byref(n)),
                     "circuit_prove")
        return out[: n.value]

    def verify(self, proof):
        """-> (status, refusing stage): (0, 0) = accepted"""
        pf = np.ascontiguousarray(proof, dtype=np.uint64)
        reason = C.c_int(0)
        rc = self.L.sipp_circuit_verify(self.h, pf.ctypes.data, len(pf), C.byref(reason))
        return rc, reason.value

    def close(self):
        if self.h:
            self.L.sipp_circuit_destroy(self.h)
            self.h = None


def io_shard(num_io, world, rank):
    """sipp_io_shard: the contiguous range (first, count) of an obligation list that `rank` of `world` proves"""
    first, count = C.c_size_t(), C.c_size_t()
    rc = lib().sipp_io_shard(int(num_io), int(world), int(rank), C.byref(first), C.byref(count))
    if rc != 0:
        raise SippError(rc, "io_shard")
    return first.value, count.value


def shard_ios(ios, world, rank):
    """IO-sharded sub-proofs (level L-D): the slices of the three obligation lists of ONE SIPP instance
    that `rank` proves as STARKs of their own; no data is exchanged between the ranks"""
    out = []
    for a in ios:
        first, count = io_shard(a.shape[0], world, rank)
        out.append(a[first: first + count])
    return out
