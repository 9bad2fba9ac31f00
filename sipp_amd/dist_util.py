"""Multi-process helpers for bench.py: the hot path shards by independent SIPP instances (one process per GPU,
no data-path collective); torch.distributed is used only for the barrier and the max-over-ranks step time."""
import os


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


def whole_job_rate(units_per_rank, world, max_step_seconds):
    """aggregate throughput: every rank processed `units_per_rank` units in (at most) max_step_seconds"""
    return world * units_per_rank / max_step_seconds
