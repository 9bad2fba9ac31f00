#!/bin/bash
# where the waves of the transform kernels spend their time (SQ counters, quad-cycle units): pmc_tree.sh <log_n> <columns>
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_tree_$1
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d "$OUT/sq" -o run -- python3 "$R/scripts/perf_generic.py" $1 $2 > "$OUT/sq.txt" 2> "$OUT/sq.log"
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_LEVEL_VMEM --output-format csv -d "$OUT/sq2" -o run -- python3 "$R/scripts/perf_generic.py" $1 $2 > "$OUT/sq2.txt" 2> "$OUT/sq2.log"
python3 - "$OUT" <<'PY'
import csv, collections, sys, re
out = sys.argv[1]
for tag in ("sq", "sq2"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    try:
        rows = csv.DictReader(open("%s/%s/run_counter_collection.csv" % (out, tag)))
    except FileNotFoundError:
        print(tag, "missing"); continue
    for r in rows:
        k = re.split(r"[<(]", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k, c in agg.items():
        if "tree" in k or "poseidon_leaves" in k:
            print(tag, k, len(n[k]), {a: "%.3g" % b for a, b in sorted(c.items())})
PY
