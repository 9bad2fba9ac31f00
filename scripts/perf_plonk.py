#!/usr/bin/env python3
"""The `outer_plonk` leg of bench.py alone (for rocprofv3 --kernel-trace --stats): plonky2's whole prove() -- witness generation level by
level on the device, then everything below it -- at the standard_ecc_config shape on the chained synthetic circuit with gates, generators
and the level schedule as data.  usage: perf_plonk.py [log2 rows = 18] [steps = 5]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 18
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
print(json.dumps(bench.outer_plonk_leg(0, log_n, steps)))
