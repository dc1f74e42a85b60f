#!/bin/bash
# Second diagnostic PMC set for the bench kernel: instruction cache, MFMA pipe and LDS (which CU-level resource do the
# eight resident waves of a CU contend for?).  Same protocol as tools/pmc_run2.sh: separate --kernel-trace --pmc passes.
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${DOMPC_PMC_DIR:-pmc3}
mkdir -p $OUT
CMD="python $R/bench.py --steps 1 --warmup 0 --batch ${DOMPC_PMC_BATCH:-4096} --no-cpu-baseline --no-traffic --no-b1"
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  [ $i -gt ${DOMPC_PMC_PASSES:-99} ] && break
  timeout -k 5 ${DOMPC_PMC_TIMEOUT:-100} rocprofv3 --kernel-trace --pmc $line --output-format csv -d $OUT/q$i -- $CMD > $OUT/q$i.log 2>&1
done <<'PASSES'
SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES
SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INST_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_CYCLES SQ_INSTS_SMEM
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL
PASSES
python - "$OUT" <<'PY'
import collections, csv, glob, json, os, sys
out = sys.argv[1]
tot = {}
for d in sorted(glob.glob(os.path.join(out, "q*"))):
    if not os.path.isdir(d): continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[:1]:
        agg = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if "dompc_solve" in r.get("Kernel_Name", ""):
                agg[r["Counter_Name"]] += float(r["Counter_Value"])
        tot.update(agg)
json.dump(tot, open(os.path.join(out, "summary.json"), "w"), indent=1)
for k, v in tot.items():
    print(f"{k:36s} {v:.6g}")
PY
