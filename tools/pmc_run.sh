#!/bin/bash
# PMC passes for the bench kernel (separate runs; --kernel-trace only, as the pool requires)
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
CMD="python $R/bench.py --steps 1 --warmup 0 --batch ${DOMPC_PMC_BATCH:-1024} --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/p1 -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $OUT/p2 -- $CMD > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/p3 -- $CMD > $OUT/p3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/p4 -- $CMD > $OUT/p4.log 2>&1
for p in p1 p2 p3 p4; do
  f=$(find $OUT/$p -name "*counter_collection.csv" | head -1)
  echo "== $p $f"
  python - "$f" <<'PY'
import csv, sys, collections
f = sys.argv[1]
agg = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    if "dompc_solve" in r.get("Kernel_Name", ""):
        agg[r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items():
    print(f"{k:28s} {v:.6g}")
PY
done
