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


def per_dispatch(tag, keep_template=False):
    """{dispatch id: {"kernel", "ns", counters...}} of a PMC pass (counter rows carry the dispatch's own timestamps)"""
    d = {}
    for r in csv.DictReader(open("%s/%s/run_counter_collection.csv" % (src, tag))):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        name = re.sub(r"^void ", "", name)
        name = re.split(r"[(]", name)[0] if keep_template else kname(r["Kernel_Name"])
        e = d.setdefault(r["Dispatch_Id"], {"kernel": name, "ns": float(r["End_Timestamp"]) - float(r["Start_Timestamp"])})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return d


SIMDS = 1024.0


def alone(disp, kernel):
    """a kernel ALONE on the chip (dispatches are serialised under counter collection): launch time, the clock the chip held
    (GRBM_GUI_ACTIVE is the sum over the 8 XCDs), VALU instructions, the quad-cycles (4 cycles) the VALUs were busy with them, and
    the share of the launch's SIMD cycles that is"""
    rows = [e for e in disp.values() if e["kernel"] == kernel and e.get("SQ_INSTS_VALU", 0) > 1e6]
    if not rows:
        return None
    n = len(rows)
    ns = sum(e["ns"] for e in rows)
    gui = sum(e.get("GRBM_GUI_ACTIVE", 0.0) for e in rows)
    insts = sum(e["SQ_INSTS_VALU"] for e in rows)
    quads = sum(e.get("SQ_ACTIVE_INST_VALU", 0.0) for e in rows)
    dual = sum(e.get("SQ_ACTIVE_INST_VALU2", 0.0) for e in rows)
    clock = gui / 8.0 / ns            # cycles per ns = GHz
    return {"launches": n, "avg_launch_ms": ns / n * 1e-6, "GRBM_GUI_ACTIVE_per_launch": gui / n, "eff_clock_ghz": clock,
            "valu_insts_per_launch": insts / n, "valu_busy_quad_cycles_per_launch": quads / n, "dual_issue_quad_cycles_per_launch": dual / n,
            # gfx950: SQ_ACTIVE_INST_VALU counts one quad-cycle per instruction even where two issued together (it equals SQ_INSTS_VALU
            # in every kernel and in every form of the micro-benchmark); the quad-cycles the VALU was really busy are that MINUS
            # SQ_ACTIVE_INST_VALU2 (quad-cycles in which a pair issued) -- v_add_u32: 0.55 per instruction = 2.2 cycles, as timed
            "quad_cycles_per_inst": (quads - dual) / insts,
            "cycles_per_inst": 4.0 * (quads - dual) / insts,
            # SIMD cycles the VALUs were busy / SIMD cycles of the launch (1024 SIMDs x GRBM_GUI_ACTIVE / 8)
            "valu_busy_frac": 4.0 * (quads - dual) / (SIMDS * gui / 8.0),
            "valu_issue_slots_frac": 4.0 * quads / (SIMDS * gui / 8.0),
            "wave_cycles_per_launch": sum(e.get("SQ_WAVE_CYCLES", 0.0) for e in rows) / n,
            "wait_inst_any_per_launch": sum(e.get("SQ_WAIT_INST_ANY", 0.0) for e in rows) / n}


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


vd = per_dispatch("pmc_v")
line = json.loads(open("%s/bench_line.json" % src).read().strip().splitlines()[-1])
# calibration: the same counters over scripts/ubench/enc_rates (one kernel per instruction form, second launch of each)
calib = None
try:
    ed = per_dispatch("pmc_enc", keep_template=True)
    forms = [l.split()[0] + (" " + l.split()[1] if l.split()[1].startswith("(") else "") for l in open("%s/enc_rates.txt" % src) if " ms " in l]
    seen, order = {}, []
    for did in sorted(ed, key=lambda x: int(x)):
        e = ed[did]
        if e["kernel"] not in seen:
            order.append(e["kernel"])
        seen[e["kernel"]] = e                      # the later (warm) launch of a form wins
    calib = []
    for i, kname_ in enumerate(order):
        e = seen[kname_]
        if e.get("SQ_INSTS_VALU", 0) < 1e6:
            continue
        gui = e.get("GRBM_GUI_ACTIVE", 0.0)
        calib.append({"form": forms[len(calib)] if len(calib) < len(forms) else kname_, "kernel": kname_,
                      "cycles_per_inst": 4.0 * (e.get("SQ_ACTIVE_INST_VALU", 0.0) - e.get("SQ_ACTIVE_INST_VALU2", 0.0)) / e["SQ_INSTS_VALU"],
                      "insts_issued_in_pairs": 2.0 * e.get("SQ_ACTIVE_INST_VALU2", 0.0) / e["SQ_INSTS_VALU"],
                      "eff_clock_ghz": gui / 8.0 / e["ns"],
                      # the share of the launch's SIMD cycles the VALUs were busy: what a pure instruction stream of this form reaches
                      # (8 waves per SIMD, eight independent registers) -- the PRACTICAL ceiling of `valu_busy_frac`
                      "valu_busy_frac": 4.0 * (e.get("SQ_ACTIVE_INST_VALU", 0.0) - e.get("SQ_ACTIVE_INST_VALU2", 0.0)) / (SIMDS * gui / 8.0) if gui else None})
    shutil.copy("%s/enc_rates.txt" % src, dst + "_enc_rates.txt")
    with open(dst + "_enc_rates.txt", "a") as fh:
        fh.write("\n# scripts/ubench/bin/canon_rates\n" + open("%s/canon_rates.txt" % src).read())
    if not calib or not forms:
        raise RuntimeError("empty calibration (enc_rates.txt / pmc_enc without usable rows)")
except Exception as ex:            # noqa: BLE001
    # no calibrated fields without a calibration: bench.py's `peak_calibrated` figures rest on these rows
    print("no enc_rates calibration:", ex)
    calib = None
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
    # which code these counters belong to: bench.py refuses them for any other (sipp_amd/build.py source_hash)
    "source_sha256": open("%s/source_sha256.txt" % src).read().strip(),
    "valu_pass_command": "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --inflight 1",
    # every kernel alone on the chip (counter collection serialises the dispatches)
    "leaf_one_alone": alone(vd, "poseidon_leaves_kernel"),
    "leaf_pair_alone": alone(vd, "poseidon_leaves_pair_kernel"),
    "transforms_alone": {k: alone(vd, k) for k in ("tree_fwd_dma_kernel", "tree_pass_dma_kernel", "tree_pass_kernel", "tree_mid_kernel", "tree_gather_kernel", "lde_column_kernel")
                         if alone(vd, k)},
    "valu_calibration": {"what": "the same counters over scripts/ubench/enc_rates (profiles/<round>_enc_rates.txt): quad-cycles (4 cycles) the VALU "
                                 "is busy per instruction of each FORM; a form that pairs (SQ_ACTIVE_INST_VALU2) costs half a quad-cycle",
                         "forms": calib},
}
NTT = ("ntt_pass_kernel", "lde_column_kernel", "bitrev_tiled_kernel", "bitrev_cols_kernel", "tree_gather_kernel", "tree_mid_kernel",
       "tree_pass_kernel", "tree_pass_dma_kernel", "tree_fwd_dma_kernel")
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
try:      # the outer-prover leg (scripts/perf_plonk.py under rocprofv3 --kernel-trace --stats)
    shutil.copy("%s/plonk/run_kernel_stats.csv" % src, dst + "_plonk_kernel_stats.csv")
    shutil.copy("%s/plonk.txt" % src, dst + "_plonk.txt")
except Exception as ex:            # noqa: BLE001
    print("no outer-prover profile:", ex)
# the side legs of scripts/profile_round.sh: the native chain, messages -> G2, plain against hardened, the final-pairing STARK (round 6)
for sub, name in (("native", "native_chain_n128"), ("mapg2", "mapg2"), ("hardened", "hardened"), ("pairing", "pairing")):
    try:
        shutil.copy("%s/%s/run_kernel_stats.csv" % (src, sub), "%s_%s_kernel_stats.csv" % (dst, name))
        shutil.copy("%s/%s.txt" % (src, sub), "%s_%s.txt" % (dst, name))
    except Exception as ex:        # noqa: BLE001
        print("no %s profile:" % sub, ex)
try:      # stream concurrency of the last traced step (scripts/timeline.py over the kernel trace of the timeline pass)
    import subprocess, os
    tl = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "timeline.py"), "%s/tl/run_kernel_trace.csv" % src],
                        capture_output=True, text=True, timeout=600)
    if tl.returncode == 0 and tl.stdout.strip():
        open(dst + "_timeline.txt", "w").write(tl.stdout)
    else:
        print("no timeline:", tl.stderr[-300:])
except Exception as ex:            # noqa: BLE001
    print("no timeline:", ex)
print(json.dumps(out, indent=1)[:3000])
