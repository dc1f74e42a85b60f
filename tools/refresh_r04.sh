# The measured artefacts of the end of round 4 in one GPU call: the driver's bench command, rocprofv3 kernel statistics of the same launches,
# phase counters, the PMC passes of the solve and of the sweep-only kernel.   bash tools/refresh_r04.sh [tag]   -> gpurun_out/<tag>/
R=${1:-r04d}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$R
mkdir -p $O
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
cut -c1-300 $O/bench.json
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-b1 --no-variant-b --sweep-steps 0 > $GRAFT_REPO_ROOT/$O/stats.log 2>&1)
ls -t $O/stats/*/*kernel_stats.csv | head -1 | xargs cat | head -5
(timeout 200 python tools/gpu_profile.py industrial_poly 16384; timeout 100 python tools/gpu_profile.py industrial_poly 256; timeout 100 python tools/gpu_profile.py industrial_poly 1) > $O/phase.txt 2>&1
DOMPC_PMC_BATCH=16384 DOMPC_PMC_DIR=$R/pmc DOMPC_PMC_TIMEOUT=120 timeout 600 bash tools/pmc_run2.sh > $O/pmc.log 2>&1
tail -34 $O/pmc.log
timeout 300 bash tools/pmc_sweep.sh > $O/pmc_sweep.log 2>&1
grep "slots     0" $O/pmc_sweep.log | head -30
