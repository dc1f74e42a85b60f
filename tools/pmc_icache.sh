cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_ic
mkdir -p $OUT
rocprofv3 --list-avail 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*\|SQC_INST[A-Z_]*" | sort -u | tr '\n' ' ' > $OUT/avail.txt
cat $OUT/avail.txt
CMD="python $R/bench.py --steps 1 --warmup 0 --batch ${DOMPC_PMC_BATCH:-4096} --no-cpu-baseline --no-traffic --no-b1 --no-variant-b --sweep-steps 0"
timeout 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/p5 -- $CMD > $OUT/p5.log 2>&1
f=$(find $OUT/p5 -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    if "dompc_solve" in r.get("Kernel_Name", ""):
        agg[r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items(): print(f"{k:28s} {v:.6g}")
PY
tail -3 $OUT/p5.log
