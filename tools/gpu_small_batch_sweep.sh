# small batches of the shipped problem (cold, host buffers): the default launch shape (K workgroups per problem, groups of problems on ONE XCD)
# against the whole-chip placement and 128-thread workgroups
for B in 8 64; do
  echo "B=$B default"; python tools/gpu_time_case.py industrial_poly '{}' $B 2>&1 | tail -1 | cut -c1-100
  for cfg in "1 256 4" "1 128 4" "1 128 8" "1 256 8" "1 128 16" "1 256 16" "0 128 8" "0 128 16" "0 128 32"; do set -- $cfg
    DOMPC_WIDE_SPREAD=$1 DOMPC_WIDE_BLOCK=$2 DOMPC_WIDE=$3 python tools/gpu_time_case.py industrial_poly '{}' $B 2>&1 | tail -1 | cut -c1-100 | sed "s/^/spread=$1 block=$2 K=$3 /"; done
done
