R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_hash
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_I8 --output-format csv -d $OUT/a -o run -- python3 $R/scripts/perf_generic.py 13 4942 > $OUT/a.txt 2> $OUT/a.log
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_I8 --output-format csv -d $OUT/b -o run -- python3 $R/scripts/perf_generic.py 17 1024 > $OUT/b.txt 2> $OUT/b.log
python3 - <<PY
import csv,collections,re
for tag in "ab":
    agg=collections.defaultdict(lambda:collections.defaultdict(float)); n=collections.defaultdict(set)
    try: rows=list(csv.DictReader(open("$OUT/%s/run_counter_collection.csv"%tag)))
    except Exception as e: print(tag,"no counters",e); continue
    for r in rows:
        k=re.split(r"[(<]",r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ",""))[0]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k,v in agg.items():
        if "poseidon_leaves" in k or "merkle" in k: print(tag,k,len(n[k]),{a:"%.4g"%b for a,b in sorted(v.items())})
PY
tail -3 $OUT/a.log
