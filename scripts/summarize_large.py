#!/usr/bin/env python3
"""Summaries of scripts/profile_large.sh (the LARGE single-GPU configs) for profiles/:
usage: summarize_large.py gpurun_out/prof_<tag>_n<N> profiles/<prefix>_bench_n<N>   (writes _kernel_stats.csv, _bench_line.json, _ntt_traffic.json)"""
import collections, csv, json, re, shutil, sys

src, dst = sys.argv[1], sys.argv[2]


def kname(s):
    s = s.replace("(anonymous namespace)::", "")
    s = re.sub(r"^void ", "", s)
    return re.split(r"[<(]", s)[0]


def counters(tag):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open("%s/%s/run_counter_collection.csv" % (src, tag))):
        k = kname(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    return agg, {k: len(v) for k, v in disp.items()}


shutil.copy("%s/stats/run_kernel_stats.csv" % src, dst + "_kernel_stats.csv")
line = json.load(open("%s/bench_line.json" % src))
json.dump(line, open(dst + "_bench_line.json", "w"))
f, nf = counters("pmc_f")
w, nw = counters("pmc_w")
instances = 3      # the PMC passes run bench.py --steps 1 --warmup 1 plus the serial step bench.py appends
NTT = ("ntt_pass_kernel", "lde_column_kernel", "lde_gather_kernel", "lde_mid_kernel", "bitrev_tiled_kernel", "bitrev_cols_kernel",
       "tree_gather_kernel", "tree_mid_kernel", "tree_pass_kernel", "tree_pass_dma_kernel", "tree_fwd_dma_kernel")
alg = line["roofline_ntt"]["algorithmic_bytes_per_step"]
traffic = sum(2.0 * f[k]["FETCH_SIZE"] + w[k]["WRITE_SIZE"] for k in NTT) * 1024.0 / instances
out = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) -- python3 bench.py --n %s --steps 1 --warmup 1 --no-cpu-baseline --inflight 1"
                  % re.search(r"n(\d+)$", src).group(1),
       "instances_profiled": instances, "fetch_correction": 2.0,
       "kernels": {k: {"launches": nf.get(k, 0), "FETCH_SIZE_kb_sum": f[k]["FETCH_SIZE"], "WRITE_SIZE_kb_sum": w[k]["WRITE_SIZE"],
                       "traffic_GB_per_instance": (2.0 * f[k]["FETCH_SIZE"] + w[k]["WRITE_SIZE"]) * 1024.0 / instances / 1e9}
                   for k in sorted(set(f) | set(w), key=lambda k: -(2.0 * f[k]["FETCH_SIZE"] + w[k]["WRITE_SIZE"]))[:24]},
       "ntt": {"traffic_bytes_per_instance": traffic, "algorithmic_bytes_per_instance": alg, "traffic_over_algorithmic": traffic / alg,
               "serial_ms_per_instance": line["roofline_ntt"]["serial_ms_per_step"], "frac_of_8TBs_algorithmic": line["roofline_ntt"]["frac"],
               "frac_of_8TBs_traffic": traffic / (line["roofline_ntt"]["serial_ms_per_step"] * 1e-3) / 8e12}}
json.dump(out, open(dst + "_ntt_traffic.json", "w"), indent=1)
print(json.dumps(out["ntt"], indent=1))
