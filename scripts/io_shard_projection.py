#!/usr/bin/env python3
"""IO-sharded sub-proofs (level L-D) on ONE GPU, one shard after the other: what every rank of a
`world`-GPU run would be given, timed alone.  The ranks exchange nothing, so the slowest shard is the instance's time on
`world` GPUs up to host effects (one process per GPU, each with its own three worker threads) -- a PROJECTION from
single-GPU measurements, labelled as such, not a multi-GPU measurement (the driver's SCALE run is that).
usage: io_shard_projection.py [n=4096] [world=8] [steps=2] [hardened=1]     (hardened = 1: API kinds 4 / 5, the variant bench.py's `value`
and `io_sharded` legs prove since round 4; 0: the plain kinds)"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sipp_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
hardened = bool(int(sys.argv[4])) if len(sys.argv) > 4 else True
d = np.load(os.path.join(ROOT, "tests", "golden", "sipp_n%d_ios.npz" % n))
ios = [d["g1"], d["g2"], d["fq12"]]


def timed(lists):
    inst = sipp_amd.Instance([a.shape[0] for a in lists], hardened=hardened)
    try:
        arena = sum(int(c.workspace_bytes) for c in inst.distinct_ctxs())
        single = bool(inst.single_ctx)
        inst.prove(lists)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(steps):
            inst.prove(lists)
        inst.sync()
        return 1e3 * (time.perf_counter() - t) / steps, arena, single
    finally:
        inst.close()


whole, whole_arena, whole_single = timed(ios)
shards = []
for rank in range(world):
    mine = sipp_amd.shard_ios(ios, world, rank)
    ms, arena, single = timed(mine)
    shards.append({"rank": rank, "records": [int(a.shape[0]) for a in mine], "ms": round(ms, 2), "arena_bytes": arena, "single_ctx": single})
worst = max(s["ms"] for s in shards)
print(json.dumps({"n": n, "world": world, "projection": True, "air_variant": "hardened (kinds 4 / 5 / 2)" if hardened else "plain (kinds 0 / 1 / 2)",
                  "whole_instance_on_one_gpu_ms": round(whole, 2), "whole_instance_arena_bytes": whole_arena, "whole_instance_single_ctx": whole_single,
                  "shards_one_at_a_time": shards,
                  "slowest_shard_ms": worst, "projected_speedup": round(whole / worst, 2),
                  # the other side of the ledger (sipp_amd/proof_cost.py): what the 3 * world proofs cost a verifier, against world = 1
                  "verifier_price": {k: v for k, v in __import__("sipp_amd").proof_cost.instance_price([a.shape[0] for a in ios], world, hardened).items()},
                  "verifier_price_world1": {k: v for k, v in __import__("sipp_amd").proof_cost.instance_price([a.shape[0] for a in ios], 1, hardened).items()},
                  "note": "projection: every shard timed alone on ONE MI355X; no data moves between ranks in level L-D"}))
