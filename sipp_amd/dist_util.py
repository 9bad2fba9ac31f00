"""Multi-process helpers for bench.py: the hot path shards by independent SIPP instances (one process per GPU,
no data-path collective); torch.distributed is used only for the barrier and the max-over-ranks step time."""
import os


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_instances(n_instances, rank, world):
    """instances (independent SIPP proofs) owned by `rank`: round-robin"""
    return list(range(rank, n_instances, world))


def max_over_ranks(value, device="cpu"):
    """max of a python float over all ranks (barrier semantics included); identity when not initialised"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def min_max_over_ranks(value, device="cpu"):
    """(min, max) of a python float over all ranks: the spread of the ranks' own step times"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value), float(value)
    t = torch.tensor([value, -value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return -float(t[1].item()), float(t[0].item())


def whole_job_rate(units_per_rank, world, max_step_seconds):
    """aggregate throughput: every rank processed `units_per_rank` units in (at most) max_step_seconds"""
    return world * units_per_rank / max_step_seconds


def init_process_group(world, local_rank, rehearsal=False, single_rank_group=False):
    """one process per GPU over RCCL (backend "nccl" on ROCm); rehearsal = every rank on one GPU over gloo, the only
    difference being the backend and the device of the timing tensor.  No-op for a single rank -- unless `single_rank_group`
    asks for a group of one (a one-GPU box can then run the RCCL branch itself: barrier and the reductions on device tensors)."""
    if world <= 1 and not single_rank_group:
        return "cpu"
    import torch
    import torch.distributed as dist
    if world <= 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if rehearsal:
        dist.init_process_group("gloo")
        return "cpu"
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    return "cuda"


def barrier(sync=None):
    """rank barrier (when initialised) followed by the caller's device synchronisation"""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    if sync is not None:
        sync()


def timed_steps(step, steps, warmup, sync=None, device="cpu", before_timing=None, local_out=None):
    """bench.py's timing contract: `warmup` untimed steps, then EXACTLY `steps` steps bracketed by barrier + device sync on
    both sides; returns (max-over-ranks elapsed seconds, result of the last step).  `local_out` (a list) receives this rank's
    own time for its `steps` steps WITHOUT the closing barrier (the spread of the ranks, reported beside the contract's figure)."""
    import time
    last = None
    for _ in range(warmup):
        last = step()
    if before_timing is not None:
        before_timing()
    barrier(sync)
    t0 = time.perf_counter()
    for _ in range(steps):
        last = step()
    if local_out is not None:
        if sync is not None:
            sync()
        local_out.append(time.perf_counter() - t0)
    barrier(sync)
    elapsed = time.perf_counter() - t0
    return max_over_ranks(elapsed, device=device), last
