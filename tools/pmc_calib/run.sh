#!/bin/bash
# usage (on the GPU box): bash tools/pmc_calib/run.sh  -> gpurun_out/pmc_calib/calibration.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_calib
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/pmc_calib/stream_f64.hip -o $OUT/stream_f64 || exit 1
N=$((1<<28))
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f -- $OUT/stream_f64 $N > $OUT/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w -- $OUT/stream_f64 $N > $OUT/w.log 2>&1
python - "$OUT" "$N" <<'PY'
import csv, glob, json, os, sys
out, n = sys.argv[1], int(sys.argv[2])
res = {}
for tag, name in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
    f = glob.glob(os.path.join(out, tag, "**", "*counter_collection.csv"), recursive=True)[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "stream_f64" in r.get("Kernel_Name", "") and r["Counter_Name"] == name]
    per_launch = {}
    rows = [r for r in csv.DictReader(open(f)) if "stream_f64" in r.get("Kernel_Name", "") and r["Counter_Name"] == name]
    for r in rows:
        per_launch.setdefault(r["Dispatch_Id"], 0.0)
        per_launch[r["Dispatch_Id"]] += float(r["Counter_Value"])
    kb = sorted(per_launch.values())[len(per_launch) // 2]
    res[name + "_KB_per_launch"] = kb
true_bytes = n * 8
res["true_bytes_each_way"] = true_bytes
res["read_factor"] = true_bytes / (res["FETCH_SIZE_KB_per_launch"] * 1024.0)
res["write_factor"] = true_bytes / (res["WRITE_SIZE_KB_per_launch"] * 1024.0)
res["note"] = "bytes = counter_KB * 1024 * factor for 8 B/lane coalesced f64 traffic (stream_f64.hip, 2 GiB each way, > Infinity Cache)"
json.dump(res, open(os.path.join(out, "calibration.json"), "w"), indent=1)
print(json.dumps(res))
PY
