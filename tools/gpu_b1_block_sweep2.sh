for cfg in "128 20" "128 24" "128 28" "128 36"; do set -- $cfg; DOMPC_WIDE_BLOCK=$1 DOMPC_WIDE=$2 python tools/gpu_b1.py industrial_poly 2>&1 | grep "B=1 cold" | cut -c1-95 | sed "s/^/block=$1 K=$2 /"; done
for cfg in "128 96" "128 128" "128 160" "256 64"; do set -- $cfg; DOMPC_WIDE_BLOCK=$1 DOMPC_WIDE=$2 python bench.py --variant tree --steps 5 --warmup 2 2>/dev/null | grep '"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('tree block=$1 K=$2  ms_per_step', round(d['ms_per_step'], 2))"; done
for cfg in "128 8" "128 12" "128 16" "256 8"; do set -- $cfg; DOMPC_WIDE_BLOCK=$1 DOMPC_WIDE=$2 python tools/gpu_b1.py CSTR 2>&1 | grep "B=1 cold" | cut -c1-80 | sed "s/^/block=$1 K=$2 /"; done
