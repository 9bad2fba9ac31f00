"""world_size-2 gloo test of bench.py's N > 1 plumbing (instance sharding, barrier, max-over-ranks timing)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch.distributed as dist
from sipp_amd import dist_util as du
rank, local_rank, world = du.rank_world()
dist.init_process_group("gloo")
mine = du.shard_instances(5, rank, world)
assert mine == list(range(rank, 5, world))
dist.barrier()
t = du.max_over_ranks(1.0 + rank)          # rank 1 is the slow one
assert t == float(world), t
rate = du.whole_job_rate(128, world, t)
assert abs(rate - 128.0 * world / world) < 1e-9
# every instance is owned exactly once
import torch
owned = torch.zeros(5, dtype=torch.int64); owned[mine] = 1
dist.all_reduce(owned)
assert owned.tolist() == [1] * 5
# bench.py's timing contract (dist_util.timed_steps): W untimed + exactly K timed steps between barriers, MAX over ranks,
# whole-job value = world * units / max step time -- with a fake step whose cost depends on the rank
calls = []
def fake_step():
    calls.append(1)
    time.sleep(0.02 * (1 + rank))
    return len(calls)
marks = []
elapsed, last = du.timed_steps(fake_step, steps=3, warmup=2, sync=lambda: marks.append("sync"), device="cpu",
                               before_timing=lambda: marks.append("prof"))
assert len(calls) == 5 and last == 5, calls
assert marks == ["prof", "sync", "sync"], marks
assert 3 * 0.02 * world <= elapsed < 3 * 0.02 * world + 0.5, elapsed      # the slow rank's time on every rank
value = du.whole_job_rate(128, world, elapsed / 3)
assert abs(value - world * 128 / (elapsed / 3)) < 1e-6
assert du.init_process_group(1, 0) == "cpu"                                 # single rank: nothing to initialise
# IO-sharded sub-proofs (sipp_io_shard through the C ABI, no GPU needed): the ranks' ranges tile every obligation list exactly once
import numpy as np, sipp_amd
for n_io in (127, 14, 6, 1, 4095):
    first, count = sipp_amd.io_shard(n_io, world, rank)
    cover = torch.zeros(n_io, dtype=torch.int64); cover[first: first + count] = 1
    dist.all_reduce(cover)
    assert cover.tolist() == [1] * n_io, (n_io, cover)
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([count]))
    assert max(int(x) for x in sizes) - min(int(x) for x in sizes) <= 1      # balanced
lists = [np.arange(7 * 56, dtype=np.uint32).reshape(7, 56), np.arange(7 * 104, dtype=np.uint32).reshape(7, 104),
         np.arange(6 * 296, dtype=np.uint32).reshape(6, 296)]
mine_io = sipp_amd.shard_ios(lists, world, rank)
assert [a.shape[1] for a in mine_io] == [56, 104, 296] and sum(a.shape[0] for a in mine_io) in (9, 11)
dist.destroy_process_group()
open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "rank" + str(rank) + ".ok"), "w").write("ok")
'''


WORKER8 = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
from sipp_amd import dist_util as du
import sipp_amd
rank, local_rank, world = du.rank_world()
assert world == 8
dist.init_process_group("gloo")
# the obligation lists of BASELINE configs[3] / configs[4] cut across 8 ranks: n = 1024 -> 1023 / 1023 / 20, n = 4096 -> 4095 / 4095 / 24
for n, lists in ((1024, (1023, 1023, 20)), (4096, (4095, 4095, 24))):
    counts = []
    for n_io in lists:
        first, count = sipp_amd.io_shard(n_io, world, rank)
        cover = torch.zeros(n_io, dtype=torch.int64); cover[first: first + count] = 1
        dist.all_reduce(cover)
        assert cover.tolist() == [1] * n_io
        counts.append(count)
    allc = [torch.zeros(3, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(allc, torch.tensor(counts))
    allc = [c.tolist() for c in allc]
    if n == 1024:
        assert allc[0] == [127, 127, 2] and all(c[0] in (127, 128) and c[2] in (2, 3) for c in allc), allc
    else:
        assert allc[0] == [511, 511, 3] and all(c[0] in (511, 512) and c[2] == 3 for c in allc), allc
# the io_sharded leg's timing: max over ranks by the contract, (min, max) of the ranks' own times beside it
own = []
elapsed, _ = du.timed_steps(lambda: time.sleep(0.01 * (1 + rank)), steps=2, warmup=1, device="cpu", local_out=own)
lo, hi = du.min_max_over_ranks(own[0])
assert 2 * 0.01 <= lo < 2 * 0.01 + 0.2 and 2 * 0.08 <= hi <= elapsed + 1e-9, (lo, hi, elapsed)
assert du.min_max_over_ranks(float(rank)) == (0.0, 7.0)
dist.destroy_process_group()
open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "rank" + str(rank) + ".ok"), "w").write("ok")
'''


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)],
                         capture_output=True, text=True, timeout=180, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    # per-rank files, not stdout: the two ranks' prints interleave character by character
    assert (tmp_path / "rank0.ok").exists() and (tmp_path / "rank1.ok").exists(), out.stdout + out.stderr


def test_eight_rank_gloo_shards_of_the_large_configs(tmp_path):
    """the 8-rank launch of bench.py's io_sharded leg without the GPU: tiling and balance of the n = 1024 / n = 4096 lists over
    8 ranks and the (min, max) spread of the ranks' step times (the GPU side of the same shards: tests/test_gpu_multi.py)"""
    script = tmp_path / "worker8.py"
    script.write_text(WORKER8 % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8",
                          "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert all((tmp_path / ("rank%d.ok" % r)).exists() for r in range(8)), out.stdout + out.stderr


def test_bare_bench_builds_the_launcher_command_line(monkeypatch):
    """`python bench.py --gpus N` with no launcher starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py <same flags>` as a child, relays its stdout and returns its exit code"""
    import io
    import importlib
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    seen = {}

    class FakeProc:
        def __init__(self, cmd, env=None, cwd=None, stdout=None, text=None):
            seen.update(cmd=cmd, env=env, cwd=cwd)
            self.stdout = io.StringIO('{"n_gpus": 8}\n')

        def wait(self):
            return 3

    monkeypatch.setattr(subprocess, "Popen", FakeProc)
    monkeypatch.setenv("RANK", "5")
    rc = bench.launch_ranks(8, ["--gpus", "8", "--steps", "4", "--warmup", "1"])
    assert rc == 3
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    assert cmd[-7:] == [os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "1"]
    assert seen["env"]["MASTER_ADDR"] == "127.0.0.1" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and "RANK" not in seen["env"]


def test_bare_bench_without_a_gpu_passes_the_ranks_failure_on():
    """no GPU in this container: the two ranks the bare invocation starts refuse to run (no CPU fallback), the parent relays that
    as a non-zero exit code and prints no JSON line -- and it is the CHILDREN that were asked, i.e. the launch happened"""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU box: tests/test_gpu_multi.py::test_bare_bench_starts_its_own_ranks runs the real thing")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0
    assert "needs a GPU" in out.stderr and out.stderr.count("needs a GPU") >= 2
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
