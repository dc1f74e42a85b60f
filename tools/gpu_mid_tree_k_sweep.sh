# mid-size trees, one problem alone: 27 leaves (n_robust = 3, 516 edges) and 81 leaves (n_robust = 4, 1 452 edges) - the launch-shape rule between its two calibration points
for nr in 3 4; do
  echo "n_robust=$nr default"; python tools/gpu_time_case.py industrial_poly "{\"n_robust\":$nr,\"uncertainty\":\"paired\"}" 1 2>&1 | tail -1 | cut -c1-110
  for cfg in "256 16" "256 24" "256 32" "256 48" "128 32" "128 48" "128 64"; do set -- $cfg; DOMPC_WIDE_BLOCK=$1 DOMPC_WIDE=$2 python tools/gpu_time_case.py industrial_poly "{\"n_robust\":$nr,\"uncertainty\":\"paired\"}" 1 2>&1 | tail -1 | cut -c1-110 | sed "s/^/block=$1 K=$2 /"; done
done
