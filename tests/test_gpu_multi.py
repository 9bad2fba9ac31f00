"""The N > 1 legs of bench.py and multi-device instances (L-A replicas, L-B one ctx per sub-proof).

The driver's 8-GPU scaling run is not ours to launch; what can be exercised on a one-GPU box is exercised here:
 * bench.py --gpus 2 under torch.distributed.run in rehearsal mode (both ranks on GPU 0, gloo instead of RCCL -- the
   only two differences from the driver's launch, see sipp_amd/dist_util.init_process_group);
 * Instance(devices=...) over every visible device (skipped on a one-GPU box): per-device constant tables, arenas,
   streams; proofs must equal the single-device ones word for word."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_bench(extra_env, nproc, args, timeout=900):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + args
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


@pytest.mark.timeout(1200)
def test_bench_two_ranks_rehearsal_prints_one_whole_job_line():
    out = _run_bench({"SIPP_BENCH_REHEARSAL": "1", "SIPP_BENCH_IO_SHARD_N": "128"}, 2,
                     ["--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--inflight", "1"])
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-3000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["warmup"] == 1 and r["scaling"] == "weak"
    assert r["unit"] == "pairings/s" and r["higher_is_better"] is True
    # whole-job aggregate: both ranks proved one n = 128 instance per step, timed by the slower rank
    assert abs(r["value"] - 2 * 128 / (r["ms_per_step"] * 1e-3)) / r["value"] < 1e-6
    assert r["proof_words"] and all(w > 0 for w in r["proof_words"])
    assert "independent SIPP instance" in r["config"]["parallelism"]
    # the IO-sharded leg: one n = 128 instance cut into two ranges of obligations (63 + 64 G1 / G2 records, 7 + 7 Fq12)
    sh = r["io_sharded"]["n=128"]
    assert sh["ranks"] == 2 and sh["scaling"] == "strong" and sh["records_of_rank0"] == [63, 63, 7]
    assert abs(sh["value"] - 128 / (sh["ms_per_instance"] * 1e-3)) / sh["value"] < 1e-6
    lo, hi = sh["rank_ms_per_instance_min_max"]
    assert 0 < lo <= hi <= sh["ms_per_instance"] * 1.001
    assert r["pipelined"] is None              # more than one rank: the queue leg (5 x 3 worker threads per rank) is off by default


@pytest.mark.timeout(1500)
def test_bench_four_ranks_rehearsal_shards_the_n1024_instance():
    """BASELINE configs[3] (n = 1024 sharded across GPUs) through the driver's own command line, as far as a one-GPU box
    allows: four ranks on GPU 0 over gloo (the pool's process guard allows at most 6 processes on a card, so the 8-rank
    launch itself cannot be rehearsed here; the 8-rank SHARDS are proved one by one in
    test_world8_shards_of_the_large_configs and the 8-rank tiling / timing contract runs on the CPU in tests/test_dist_cpu.py).
    No --inflight flag: the default must switch the queue leg off for world > 1."""
    out = _run_bench({"SIPP_BENCH_REHEARSAL": "1", "SIPP_BENCH_IO_SHARD_N": "1024"}, 4,
                     ["--gpus", "4", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], timeout=1400)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-3000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 4 and r["steps"] == 2 and r["scaling"] == "weak" and r["pipelined"] is None
    assert abs(r["value"] - 4 * 128 / (r["ms_per_step"] * 1e-3)) / r["value"] < 1e-6
    sh = r["io_sharded"]["n=1024"]
    assert sh["ranks"] == 4 and sh["scaling"] == "strong" and sh["records_of_rank0"] == [255, 255, 5]
    assert abs(sh["value"] - 1024 / (sh["ms_per_instance"] * 1e-3)) / sh["value"] < 1e-6
    lo, hi = sh["rank_ms_per_instance_min_max"]
    assert 0 < lo <= hi <= sh["ms_per_instance"] * 1.001


_SHARDS_1024 = {0: [127, 127, 2], 1: [128, 128, 3], 2: [128, 128, 2], 3: [128, 128, 3], 4: [128, 128, 2], 5: [128, 128, 3], 6: [128, 128, 2],
                7: [128, 128, 3]}
_SHARDS_4096 = {r: [511 if r == 0 else 512, 511 if r == 0 else 512, 3] for r in range(8)}


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("n,log_n,records,hardened", [
    (1024, 16, _SHARDS_1024, False), (1024, 16, _SHARDS_1024, True),
    (4096, 18, {r: _SHARDS_4096[r] for r in (0, 3, 7)}, False), (4096, 18, _SHARDS_4096, True)])
def test_world8_shards_of_the_large_configs(n, log_n, records, hardened):
    """BASELINE configs[3] / configs[4]: the shards an 8-rank run proves -- n = 1024: 127 / 128 G1 and G2 records (N = 2^16)
    and 2 / 3 Fq12 records (the two-IO-block minimum and a padded block); n = 4096: 511 / 512 records, N = 2^18, the first
    size outside the fused LDE kernels' range, with ragged 511-record lists padded by one copy.  Through sipp_instance_prove exactly
    as bench.py's io_sharded leg does, for BOTH AIR variants: ALL EIGHT ranks of n = 1024 (plain and hardened) and of n = 4096 with
    the hardened kinds 4 / 5 (the variant the leg times); ranks 0, 3 and 7 of n = 4096 with the plain kinds.  The oracle's verifier
    accepts every proof, the public inputs are the rank's slice (padding = copies of its last record), and a single ctx gives the
    same words."""
    import sipp_amd
    from tests import _oracle, _verify
    d = np.load(os.path.join(ROOT, "tests", "golden", "sipp_n%d_ios.npz" % n))
    ios = [d["g1"], d["g2"], d["fq12"]]
    L = sipp_amd.lib()
    seen = [0, 0, 0]
    for rank, want_counts in records.items():
        mine = sipp_amd.shard_ios(ios, 8, rank)
        assert [int(a.shape[0]) for a in mine] == want_counts, rank
        inst = sipp_amd.Instance([a.shape[0] for a in mine], hardened=hardened)
        try:
            proofs = [p.copy() for p in inst.prove(mine)]
        finally:
            inst.close()
        for k in range(3):
            first, count = sipp_amd.io_shard(ios[k].shape[0], 8, rank)
            assert (mine[k] == ios[k][first: first + count]).all()
            seen[k] += count
            pf = proofs[k]
            kind = k + 4 if hardened and k < 2 else k
            assert int(pf[1]) == kind
            nio = int(pf[3])
            assert nio >= max(2, count) and nio & (nio - 1) == 0
            if k < 2:
                assert int(pf[2]) == log_n and nio == (1 << (log_n - 9))
            assert _verify.both_accept(pf), (rank, k)
            if n == 4096 and hardened and rank == 3:
                # BASELINE configs[4]: rank 3's world-8 shard word for word -- the sha256 of the oracle's proof of the same slice
                # (tools/gen_golden.py digests_large), not only a verifier pass
                import hashlib
                import json
                gold = json.load(open(os.path.join(ROOT, "tests", "golden", "proof_digests_large.json")))
                g = gold["n4096_world8_rank3." + ("g1_hardened", "g2_hardened", "fq12")[k]]
                assert g["records"] == count and len(pf) == g["words"] and hashlib.sha256(pf.tobytes()).hexdigest() == g["sha256"], (rank, k)
            pis = pf[-nio * mine[k].shape[1]:].reshape(nio, mine[k].shape[1])
            assert (pis[:count] == mine[k]).all() and (pis[count:] == mine[k][-1]).all(), (rank, k)
            if rank == 3:
                c = sipp_amd.Ctx(workspace_bytes=L.sipp_workspace_bytes(kind, count))
                try:
                    if hardened:
                        c._ck(L.sipp_ctx_set_hardened(c.h, 1), "set_hardened")
                    alone = c.prove(k, mine[k])
                finally:
                    c.close()
                assert len(alone) == len(pf) and (alone == pf).all(), (rank, k)
    if len(records) == 8:
        assert seen == [a.shape[0] for a in ios]          # the eight ranges tile the three lists


def test_bench_refuses_a_launcher_with_the_wrong_world_size():
    """a launcher that hands `bench.py --gpus 2` a WORLD_SIZE of 1 must not get a one-GPU number reported as the 2-GPU leg"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK")}
    env["WORLD_SIZE"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0
    assert "torch.distributed.run" in (out.stdout + out.stderr)
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


@pytest.mark.timeout(1200)
def test_bare_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher (how the driver invokes `--gpus 1`): the process starts the two ranks itself
    before it touches the GPU, relays rank 0's one line and its exit code.  Rehearsal mode = both ranks on GPU 0 over gloo."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(SIPP_BENCH_REHEARSAL="1", SIPP_BENCH_IO_SHARD_N="128", SIPP_BENCH_OTHER_AIR="0", SIPP_BENCH_MAP_G2="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=1100, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-3000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["scaling"] == "weak"
    assert abs(r["value"] - 2 * 128 / (r["ms_per_step"] * 1e-3)) / r["value"] < 1e-6
    assert r["io_sharded"]["n=128"]["ranks"] == 2


@pytest.mark.timeout(900)
def test_bench_timing_contract_over_rccl_with_one_rank():
    """the `nccl` (= RCCL) branch of dist_util.init_process_group has never had more than this box's one GPU: run it with a
    process group of ONE rank -- bench.py's barrier, max-over-ranks and (min, max)-over-ranks reductions on DEVICE tensors
    through RCCL, around the real timed region and the io_sharded leg"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(SIPP_BENCH_SINGLE_RANK_GROUP="1", SIPP_BENCH_IO_SHARD_N="128", SIPP_BENCH_OTHER_AIR="0", SIPP_BENCH_MAP_G2="0", SIPP_BENCH_OUTER_PLONK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline", "--inflight", "1"], capture_output=True, text=True, timeout=800, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert r["n_gpus"] == 1 and r["timing_reduction"] == "nccl:cuda"
    assert abs(r["value"] - 128 / (r["ms_per_step"] * 1e-3)) / r["value"] < 1e-6
    lo, hi = r["io_sharded"]["n=128"]["rank_ms_per_instance_min_max"]
    assert 0 < lo <= hi


def test_dist_util_reductions_on_device_tensors_over_rccl():
    """dist_util's helpers themselves over a one-rank RCCL group, in a child process (a process group is process state)"""
    code = ("import os, torch\n"
            "from sipp_amd import dist_util as du\n"
            "torch.cuda.set_device(0)\n"
            "dev = du.init_process_group(1, 0, single_rank_group=True)\n"
            "import torch.distributed as dist\n"
            "assert dev == 'cuda' and dist.get_backend() == 'nccl' and dist.get_world_size() == 1\n"
            "du.barrier(torch.cuda.synchronize)\n"
            "assert du.max_over_ranks(1.25, device=dev) == 1.25\n"
            "assert du.min_max_over_ranks(2.5, device=dev) == (2.5, 2.5)\n"
            "t, last = du.timed_steps(lambda: 7, 3, 1, sync=torch.cuda.synchronize, device=dev)\n"
            "assert last == 7 and t > 0\n"
            "dist.destroy_process_group()\n"
            "print('rccl one-rank ok')\n")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0 and "rccl one-rank ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_instance_over_all_visible_devices_matches_single_device():
    import torch
    import sipp_amd
    nd = torch.cuda.device_count()
    if nd < 2:
        pytest.skip("one visible GPU: multi-device ctxs cannot be exercised on this box")
    d = np.load(os.path.join(ROOT, "tests", "golden", "sipp_n8_ios.npz"))
    ios = [d["g1"], d["g2"], d["fq12"]]
    ref = sipp_amd.Instance([a.shape[0] for a in ios], devices=(0, 0, 0))
    try:
        want = [p.copy() for p in ref.prove(ios)]
    finally:
        ref.close()
    # every rotation of the visible devices over the three sub-proofs: each device gets each kind at least once
    for start in range(nd):
        devs = tuple((start + k) % nd for k in range(3))
        inst = sipp_amd.Instance([a.shape[0] for a in ios], devices=devs)
        try:
            for _ in range(2):
                got = inst.prove(ios)
                for k in range(3):
                    assert len(got[k]) == len(want[k]) and (got[k] == want[k]).all(), (devs, k)
        finally:
            inst.close()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_io_sharded_sub_proofs_cover_the_instance_and_verify(world):
    """Level L-D: the obligation lists of ONE instance cut into `world` ranges, every range proved as STARKs of its own
    (reference side: one g1_exp_circuit / g2_exp_circuit / fq12_exp_circuit call per range, src/verifier_circuit.rs:133-135).
    Every shard's proofs are what a single ctx produces for the same slice, the CPU verifier accepts them, their public
    inputs are exactly the slice, and the slices tile the lists.  world = 8 leaves ranks without any Fq12 record."""
    import sipp_amd
    from tests import _oracle, _verify
    d = np.load(os.path.join(ROOT, "tests", "golden", "sipp_n8_ios.npz"))
    ios = [d["g1"], d["g2"], d["fq12"]]
    seen = [0, 0, 0]
    ctx = sipp_amd.Ctx(workspace_bytes=max(sipp_amd.lib().sipp_workspace_bytes(k, 8) for k in range(3)))
    try:
        for rank in range(world):
            mine = sipp_amd.shard_ios(ios, world, rank)
            inst = sipp_amd.Instance([a.shape[0] for a in mine])
            try:
                proofs = [p.copy() for p in inst.prove(mine)]
            finally:
                inst.close()
            for k in range(3):
                if mine[k].shape[0] == 0:
                    assert len(proofs[k]) == 0
                    continue
                first, count = sipp_amd.io_shard(ios[k].shape[0], world, rank)
                assert first == seen[k] and (mine[k] == ios[k][first: first + count]).all()
                seen[k] += count
                alone = ctx.prove(k, mine[k])
                assert len(alone) == len(proofs[k]) and (alone == proofs[k]).all(), (rank, k)
                assert _verify.both_accept(proofs[k]), (rank, k)
    finally:
        ctx.close()
    assert seen == [a.shape[0] for a in ios]


@pytest.mark.parametrize("hardened", [False, True])
def test_one_ctx_instance_proves_back_to_back_the_same_words(hardened):
    """sipp_instance_prove with three EQUAL handles: one ctx, one arena sized for the largest proof, the three proofs one after the
    other (what sipp_amd.Instance picks by itself when three arenas would not fit the card: n = 4096 on one GPU) -- the same words
    as three ctxs on three streams; two equal handles and a third are refused"""
    import ctypes as C
    import sipp_amd
    d = np.load(os.path.join(ROOT, "tests", "golden", "sipp_n8_ios.npz"))
    ios = [d["g1"], d["g2"], d["fq12"]]
    three = sipp_amd.Instance([a.shape[0] for a in ios], hardened=hardened, single_ctx=False)
    one = sipp_amd.Instance([a.shape[0] for a in ios], hardened=hardened, single_ctx=True)
    try:
        assert not three.single_ctx and len(three.distinct_ctxs()) == 3
        assert one.single_ctx and len(one.distinct_ctxs()) == 1 and one.ctxs[0] is one.ctxs[2]
        want = [p.copy() for p in three.prove(ios)]
        for _ in range(2):
            got = one.prove(ios)
            for k in range(3):
                assert int(got[k][1]) == (k + 4 if hardened and k < 2 else k)
                assert len(got[k]) == len(want[k]) and (got[k] == want[k]).all(), k
        # the automatic choice: three arenas of this size are nowhere near 90 % of the card (Instance.SINGLE_CTX_SHARE)
        auto = sipp_amd.Instance([a.shape[0] for a in ios], hardened=hardened)
        assert not auto.single_ctx
        auto.close()
        L, vp = sipp_amd.lib(), C.c_void_p
        h = (vp * 3)(three.ctxs[0].h, three.ctxs[0].h, three.ctxs[2].h)
        pi = (vp * 3)(*[a.ctypes.data for a in ios])
        ni = (C.c_size_t * 3)(*[a.shape[0] for a in ios])
        po = (vp * 3)(*[o.ctypes.data for o in three.out])
        pc = (C.c_size_t * 3)(*three.caps)
        pl = (C.c_size_t * 3)()
        assert L.sipp_instance_prove(h, pi, ni, po, pc, pl) == -1          # SIPP_E_BADARG
    finally:
        three.close()
        one.close()


def test_device_memory_reports_the_card():
    import ctypes as C
    import sipp_amd
    L = sipp_amd.lib()
    fr, tot = C.c_size_t(), C.c_size_t()
    assert L.sipp_device_memory(0, C.byref(fr), C.byref(tot)) == 0
    assert 0 < fr.value <= tot.value and tot.value > (64 << 30)
    assert sipp_amd._lib.device_memory_bytes(0) == tot.value
    assert L.sipp_device_memory(99, None, None) != 0


def test_pool_streams_and_dedicated_queues_give_the_same_proofs():
    """the ctx streams (sipp_amd/csrc/api.hip create_ctx_stream): a hardware queue of their own by default, the runtime's pool with
    priorities under SIPP_DEDICATED_QUEUES=0 -- the knob is read once per process, so the other setting runs in a child process;
    scheduling must not show in a proof"""
    import hashlib
    import sipp_amd
    d = np.load(os.path.join(ROOT, "tests", "golden", "sipp_n4_ios.npz"))
    ios = [d["g1"], d["g2"], d["fq12"]]
    inst = sipp_amd.Instance([a.shape[0] for a in ios])
    try:
        here = [hashlib.sha256(p.tobytes()).hexdigest() for p in inst.prove(ios)]
    finally:
        inst.close()
    code = ("import hashlib, numpy as np, sipp_amd\n"
            "d = np.load('tests/golden/sipp_n4_ios.npz'); ios = [d['g1'], d['g2'], d['fq12']]\n"
            "inst = sipp_amd.Instance([a.shape[0] for a in ios])\n"
            "print(' '.join(hashlib.sha256(p.tobytes()).hexdigest() for p in inst.prove(ios)))\n"
            "inst.close()\n")
    env = dict(os.environ, SIPP_DEDICATED_QUEUES="0")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split()[-3:] == here


def test_instance_queue_refuses_slots_that_do_not_fit_and_never_picks_one_ctx():
    """sipp_instances_prove needs three DISTINCT ctxs per slot, so InstanceQueue never takes Instance's one-ctx fallback, and
    a queue whose arenas exceed the card's free memory is refused before the first allocation with SIPP_E_NOMEM and a message that says
    what to change"""
    import sipp_amd
    d = np.load(os.path.join(ROOT, "tests", "golden", "sipp_n8_ios.npz"))
    ios = [d["g1"], d["g2"], d["fq12"]]
    q = sipp_amd.InstanceQueue([a.shape[0] for a in ios], in_flight=2)
    try:
        assert all(not s.single_ctx and len(s.distinct_ctxs()) == 3 for s in q.slots)
        proofs = q.prove([ios, ios, ios])
        assert len(proofs) == 3 and all(len(p[k]) > 0 for p in proofs for k in range(3))
    finally:
        q.close()
    with pytest.raises(sipp_amd.SippError) as e:
        sipp_amd.InstanceQueue([4095, 4095, 24], in_flight=4, hardened=True)        # 4 x 276 GB
    assert e.value.code == -3 and "in_flight" in str(e.value)


def test_one_ctx_instance_attempts_every_proof_and_returns_the_first_failure():
    """the one-ctx branch of sipp_instance_prove behaves like the three-ctx path: a kind that fails (an unprovable G2 record) does not
    stop the others; the first failing status comes back and the failed kind reports length 0"""
    import ctypes as C
    import sipp_amd
    from tests import _oracle, _verify
    d = np.load(os.path.join(ROOT, "tests", "golden", "sipp_n8_ios.npz"))
    ios = [d["g1"].copy(), d["g2"].copy(), d["fq12"].copy()]
    ios[1][0, -1] ^= 1                                       # claimed output off by one
    one = sipp_amd.Instance([a.shape[0] for a in ios], single_ctx=True)
    try:
        L, vp = sipp_amd.lib(), C.c_void_p
        h = (vp * 3)(*[c.h for c in one.ctxs])
        pi = (vp * 3)(*[a.ctypes.data for a in ios])
        ni = (C.c_size_t * 3)(*[a.shape[0] for a in ios])
        po = (vp * 3)(*[o.ctypes.data for o in one.out])
        pc = (C.c_size_t * 3)(*one.caps)
        pl = (C.c_size_t * 3)()
        assert L.sipp_instance_prove(h, pi, ni, po, pc, pl) == -8          # SIPP_E_WITNESS
        assert pl[1] == 0 and pl[0] > 0 and pl[2] > 0
        assert _verify.both_accept(one.out[0][: pl[0]]) and _verify.both_accept(one.out[2][: pl[2]])
    finally:
        one.close()
