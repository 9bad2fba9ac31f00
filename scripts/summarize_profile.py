#!/usr/bin/env python3
"""Turns the rocprofv3 outputs of scripts/profile_round.sh into the small summaries committed under profiles/.
usage: summarize_profile.py gpurun_out/prof_<tag> profiles/<prefix>"""
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


shutil.copy("%s/stats/run_kernel_stats.csv" % src, dst + "_bench_n128_kernel_stats.csv")
shutil.copy("%s/bench_line.json" % src, dst + "_bench_n128_bench_line.json")
f, nf = counters("pmc_f")
w, nw = counters("pmc_w")
v, nv = counters("pmc_v")
steps_profiled = 4   # --steps 2 --warmup 1, plus the serial step bench.py appends for kernel_ms_serial
tot = sum(x["SQ_INSTS_VALU"] for x in v.values())


def leaf(k, what):
    n = max(1, nf.get(k, 0))
    return {"kernel": k, "what": what, "launches": nf.get(k, 0), "FETCH_SIZE_kb_sum": f[k]["FETCH_SIZE"], "WRITE_SIZE_kb_sum": w[k]["WRITE_SIZE"],
            "traffic_bytes_per_launch": (2.0 * f[k]["FETCH_SIZE"] + w[k]["WRITE_SIZE"]) * 1024.0 / n,
            "valu_insts_per_launch": v[k]["SQ_INSTS_VALU"] / max(1, nv.get(k, 0))}


line = json.loads(open("%s/bench_line.json" % src).read().strip().splitlines()[-1])
out = {
    # the AIR variant of the profiled instance (bench.py uses these counters only for a line of the same kinds)
    "kinds": line["config"]["kinds"],
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE, --pmc SQ_INSTS_VALU) --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --inflight 1",
    "calibration": "scripts/ubench/fetch_calib.hip: 1 GiB read with the leaf kernel's 8-B-per-lane column pattern reports FETCH_SIZE = 524,293.5 KB (exactly 1/2, as MI355X_MICROARCH.md section HBM says for wide coalesced reads); 1 GiB written reports WRITE_SIZE = 1,048,576 KB (exact)",
    "fetch_correction": 2.0,
    "instances_profiled": steps_profiled,
    "leaf_one": leaf("poseidon_leaves_kernel", "one state per lane: the trees of more than 2^16 leaves (G1 / G2 at n = 128); launches with <= 4 columns are copies (hash_or_noop) and are counted in `launches` of this kernel name only if the kernel was launched for them"),
    "leaf_pair": leaf("poseidon_leaves_pair_kernel", "two lanes per state: the thin Fq12 trees"),
    "valu_insts_per_instance": tot / steps_profiled,
}
NTT = ("ntt_pass_kernel", "lde_column_kernel", "bitrev_tiled_kernel", "bitrev_cols_kernel", "tree_gather_kernel", "tree_mid_kernel",
       "tree_pass_kernel")
out["ntt"] = {"kernels": {k: {"launches": nf.get(k, 0), "FETCH_SIZE_kb_sum": f[k]["FETCH_SIZE"], "WRITE_SIZE_kb_sum": w[k]["WRITE_SIZE"]}
                          for k in NTT if nf.get(k, 0)},
              "traffic_bytes_per_instance": sum(2.0 * f[k]["FETCH_SIZE"] + w[k]["WRITE_SIZE"] for k in NTT) * 1024.0 / steps_profiled,
              "note": "all NTT / LDE kernels of one n = 128 instance (4 instances profiled: 1 warm-up + 2 timed steps + the serial step); FETCH_SIZE x2"}
out["merkle"] = {k: nf.get(k, 0) // steps_profiled for k in ("merkle_subtree_kernel",) if nf.get(k, 0)}
json.dump({"command": "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --inflight 1",
           "instances_profiled": steps_profiled, "valu_insts_per_instance": tot / steps_profiled,
           "per_kernel": {k: {"SQ_INSTS_VALU_per_instance": x["SQ_INSTS_VALU"] / steps_profiled, "share": x["SQ_INSTS_VALU"] / tot,
                              "launches_per_instance": nv[k] / steps_profiled}
                          for k, x in sorted(v.items(), key=lambda kv: -kv[1]["SQ_INSTS_VALU"]) if x["SQ_INSTS_VALU"] > 0}},
          open(dst + "_valu_by_kernel.json", "w"), indent=1)
json.dump(out, open(dst + "_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
