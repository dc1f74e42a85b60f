#!/bin/bash
# Diagnostic PMC passes for the bench kernel (what saturates?): SQ issue mix, memory pipeline, L2 / fabric, TLB.
# Separate runs with --kernel-trace only (the pool refuses --pmc together with the API trace domains).
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${DOMPC_PMC_DIR:-pmc2}
mkdir -p $OUT
CMD="python $R/bench.py --steps 1 --warmup 0 --batch ${DOMPC_PMC_BATCH:-16384} --no-cpu-baseline --no-traffic --no-b1 --no-variant-b --sweep-steps 0"
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  [ $i -gt ${DOMPC_PMC_PASSES:-99} ] && break
  timeout -k 5 ${DOMPC_PMC_TIMEOUT:-150} rocprofv3 --kernel-trace --pmc $line --output-format csv -d $OUT/q$i -- $CMD > $OUT/q$i.log 2>&1
done <<'PASSES'
FETCH_SIZE
WRITE_SIZE
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum
GRBM_GUI_ACTIVE GRBM_EA_BUSY TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
GRBM_TA_BUSY GRBM_TC_BUSY TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_FLAT
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
