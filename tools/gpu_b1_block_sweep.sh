# one problem alone: wavefronts per workgroup of the wide mode (DOMPC_WIDE_BLOCK = 64 / 128 / 256 threads) against workgroups per problem
for cfg in "256 12" "128 16" "128 24" "128 32" "64 32" "64 48" "256 10" "256 14"; do set -- $cfg; DOMPC_WIDE_BLOCK=$1 DOMPC_WIDE=$2 python tools/gpu_b1.py industrial_poly 2>&1 | grep "B=1 cold" | cut -c1-95 | sed "s/^/block=$1 K=$2 /"; done
