# Re-creates the measured artefacts of a round in one GPU call (every step bounded by its own timeout):
#   bash tools/refresh_profiles.sh r02      -> gpurun_out/r02/...   (copy what should be judged into profiles/)
R=${1:-r02}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$R
mkdir -p $O
timeout 300 python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err
cut -c1-400 $O/bench.json
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-b1 --no-variant-b --sweep-steps 0 > $GRAFT_REPO_ROOT/$O/stats.log 2>&1)
ls -t $O/stats/*/*kernel_stats.csv | head -1 | xargs cat | head -5
# QUICK=1: the phase profile at the bench batch size only and the two traffic passes of the counters (FETCH_SIZE, WRITE_SIZE)
if [ -n "${QUICK:-}" ]; then
  (timeout 200 python tools/gpu_profile.py industrial_poly ${DOMPC_BENCH_BATCH:-16384}) > $O/phase.txt 2>&1
  export DOMPC_PMC_PASSES=2
else
  (timeout 200 python tools/gpu_profile.py industrial_poly ${DOMPC_BENCH_BATCH:-16384}; timeout 100 python tools/gpu_profile.py industrial_poly 256; timeout 100 python tools/gpu_profile.py industrial_poly 1; timeout 200 python tools/gpu_check.py 2>&1 | tail -12) > $O/phase.txt 2>&1
fi
DOMPC_PMC_BATCH=${DOMPC_BENCH_BATCH:-16384} DOMPC_PMC_DIR=$R/pmc DOMPC_PMC_TIMEOUT=120 timeout 700 bash tools/pmc_run2.sh > $O/pmc.log 2>&1
tail -32 $O/pmc.log
