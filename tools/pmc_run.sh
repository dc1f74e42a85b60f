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
python - "$OUT" "${DOMPC_PMC_BATCH:-1024}" <<'PY'
import collections, csv, glob, json, os, sys
out, batch = sys.argv[1], int(sys.argv[2])
tot = {}
for p in ("p1", "p2", "p3", "p4"):
    fs = glob.glob(os.path.join(out, p, "**", "*counter_collection.csv"), recursive=True)
    agg, rows = collections.defaultdict(float), 0
    for f in fs[:1]:
        for r in csv.DictReader(open(f)):
            if "dompc_solve" in r.get("Kernel_Name", ""):
                agg[r["Counter_Name"]] += float(r["Counter_Value"])
                rows += 1
    rec = {"pass": p, "kernel": "dompc_solve_kernel",
           "command": f"bench.py --steps 1 --warmup 0 --batch {batch} --no-cpu-baseline", "counters": dict(agg)}
    json.dump(rec, open(os.path.join(out, p + ".json"), "w"), indent=1)
    tot.update(agg)
    for k, v in agg.items():
        print(f"{p} {k:28s} {v:.6g}")
if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
    rf, wf = 2.0, 1.0      # profiles/pmc_calibration.json (tools/pmc_calib/run.sh): 8 B/lane f64 traffic on gfx950
    cal = os.path.join(os.path.dirname(os.path.dirname(out)), "profiles", "pmc_calibration.json")
    if os.path.exists(cal):
        c = json.load(open(cal)); rf, wf = round(c["read_factor"], 3), round(c["write_factor"], 3)
    summ = {"batch": batch, "variant": "A", "launches": 1, "FETCH_SIZE_KB": tot["FETCH_SIZE"], "WRITE_SIZE_KB": tot["WRITE_SIZE"],
            "read_factor": rf, "write_factor": wf,
            "hbm_bytes_per_launch_raw": (tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0,
            "hbm_bytes_per_launch": (tot["FETCH_SIZE"] * rf + tot["WRITE_SIZE"] * wf) * 1024.0,
            "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/pmc_run.sh: bench.py --steps 1 --warmup 0 "
                    "--batch N), summed over the dispatch rows of dompc_solve_kernel; counter units are KB. Corrected with the factors calibrated for "
                    "8 B/lane coalesced f64 traffic (tools/pmc_calib/): FETCH_SIZE x2.0, WRITE_SIZE x1.0 on gfx950."}
    json.dump(summ, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
    print(json.dumps(summ)[:200])
PY
