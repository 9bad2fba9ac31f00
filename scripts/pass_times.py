"""Per-launch durations of the transform kernels of ONE commit (last iteration) from a rocprofv3 kernel trace of perf_generic.py.
usage: pass_times.py <kernel_trace.csv> <launches per commit to show>"""
import csv, re, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = re.split(r"[<(]", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
keep = [r for r in rows if r[2] in ("ntt_pass_kernel", "bitrev_cols_kernel", "lde_gather_kernel", "lde_mid_kernel", "lde_column_kernel",
                                    "poseidon_leaves_kernel", "zero_pad_kernel", "scale_kernel") or "bitrev" in r[2] or "pad" in r[2]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for s, e, nm in keep[-n:]:
    print("%-24s %9.3f ms" % (nm, (e - s) / 1e6))
