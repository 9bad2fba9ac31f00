#!/bin/bash
# Where do the leaf-hash kernel's waves wait?  SQ / SQC counters of scripts/perf_generic.py 17 1024 (poseidon_leaves_kernel alone on the chip):
# instruction cache, scalar (constant) loads, LDS, MFMA.  Three passes (counter slots); GPU box.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_leaf_stalls
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -o run -- python3 $R/scripts/perf_generic.py 17 1024 > $OUT/a.txt 2> $OUT/a.log || exit 1
rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQ_INSTS_SMEM SQ_INSTS_SALU --output-format csv -d $OUT/b -o run -- python3 $R/scripts/perf_generic.py 17 1024 > $OUT/b.txt 2> $OUT/b.log || exit 1
rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 --output-format csv -d $OUT/c -o run -- python3 $R/scripts/perf_generic.py 17 1024 > $OUT/c.txt 2> $OUT/c.log || exit 1
rocprofv3 --kernel-trace --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_BUSY_CYCLES SQ_INSTS_BRANCH SQ_LEVEL_WAVES --output-format csv -d $OUT/d -o run -- python3 $R/scripts/perf_generic.py 17 1024 > $OUT/d.txt 2> $OUT/d.log || exit 1
python3 - <<PY
import csv,collections,re
for tag in "abcd":
    agg=collections.defaultdict(lambda:collections.defaultdict(float)); n=collections.defaultdict(set)
    try: rows=list(csv.DictReader(open("$OUT/%s/run_counter_collection.csv"%tag)))
    except Exception as e: print(tag,"no counters",e); continue
    for r in rows:
        k=re.split(r"[(<]",r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ",""))[0]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k,v in agg.items():
        if k.startswith("poseidon_leaves") or k.startswith("tree_fwd"): print(tag,k,len(n[k]),{a:"%.4g"%b for a,b in sorted(v.items())})
PY
