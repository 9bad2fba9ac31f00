#!/bin/bash
# per-kernel SQ counters of scripts/perf_generic.py <log_n> <cols> (GPU box): where do the waves of the transform kernels spend their cycles?
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$1_$2
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS --output-format csv -d $OUT/a -o run -- python3 $R/scripts/perf_generic.py $1 $2 > $OUT/a.txt 2> $OUT/a.log || exit 1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -o run -- python3 $R/scripts/perf_generic.py $1 $2 > $OUT/b.txt 2> $OUT/b.log || exit 1
python3 - <<PY
import csv,collections,re
def kname(s):
    s=s.replace("(anonymous namespace)::","");s=re.sub(r"^void ","",s);return re.split(r"[(]",s)[0]
for tag in "ab":
    agg=collections.defaultdict(lambda:collections.defaultdict(float)); n=collections.defaultdict(set)
    for r in csv.DictReader(open("$OUT/%s/run_counter_collection.csv"%tag)):
        k=kname(r["Kernel_Name"]); agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k,v in agg.items():
        if "tree" in k or "ntt" in k or "lde" in k or "poseidon_leaves" in k:
            print(tag,k,len(n[k]),{a:"%.4g"%b for a,b in v.items()})
PY
