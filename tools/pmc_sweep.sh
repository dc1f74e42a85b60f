#!/bin/bash
# Counters of the sweep-only kernel (tools/pmc_sweep.py): instructions and busy cycles per edge of the Jacobian sweep, at two and at
# one resident wavefront per SIMD (DOMPC_SLOTS=1024).  Separate --kernel-trace --pmc passes, counters that the box does not list are dropped.
#   gpurun -- 'bash tools/pmc_sweep.sh'   -> gpurun_out/pmc_sweep/summary_*.json
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_sweep
mkdir -p $OUT
B=${DOMPC_PMC_BATCH:-16384}
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ[C]*_[A-Z0-9_]*" | sort -u > $OUT/avail.txt
for SL in 0 1024; do
  if [ $SL -gt 0 ]; then export DOMPC_SLOTS=$SL; else unset DOMPC_SLOTS; fi
  i=0
  while read -r line; do
    [ -z "$line" ] && continue
    i=$((i+1))
    keep=""
    for c in $line; do grep -qx "$c" $OUT/avail.txt && keep="$keep $c"; done
    [ -z "$keep" ] && continue
    timeout -k 5 150 rocprofv3 --kernel-trace --pmc $keep --output-format csv -d $OUT/s${SL}_q$i -- python $R/tools/pmc_sweep.py $B 2 > $OUT/s${SL}_q$i.log 2>&1
    tail -1 $OUT/s${SL}_q$i.log
  done <<'PASSES'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_CYCLES
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY
PASSES
  python - "$OUT" "$SL" "$B" <<'PY'
import collections, csv, glob, json, os, sys
out, sl, B = sys.argv[1], sys.argv[2], int(sys.argv[3])
tot = {}
for d in sorted(glob.glob(os.path.join(out, f"s{sl}_q*"))):
    if not os.path.isdir(d): continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[:1]:
        per = collections.defaultdict(lambda: collections.defaultdict(float))     # dispatch -> counter -> value
        for r in csv.DictReader(open(f)):
            if "dompc_solve" in r.get("Kernel_Name", ""):
                per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        if per:
            last = per[sorted(per, key=int)[-1]]          # the last launch (the first one warms the caches)
            tot.update(last)
edges = B * 180
res = {"slots": sl, "batch": B, "edges_per_launch": edges, "counters_of_one_launch": tot,
       "per_edge": {k: v / edges for k, v in tot.items()}}
json.dump(res, open(os.path.join(out, f"summary_slots{sl}.json"), "w"), indent=1)
for k, v in tot.items():
    print(f"slots {sl:>5s}  {k:30s} {v:14.6g}   per edge {v / edges:10.2f}")
PY
done
