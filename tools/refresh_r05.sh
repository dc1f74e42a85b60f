# The measured artefacts of the end of round 5 in one GPU call: smoke, the driver's bench command, rocprofv3 kernel statistics of the same
# launches, phase counters (scheduler of round 5 and the default one), the PMC passes of the solve and of the sweep-only kernel, the
# 243-leaf tree.   bash tools/refresh_r05.sh [tag]   -> gpurun_out/<tag>/
R=${1:-r05g}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$R
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
grep '^{"metric' $O/bench.json | cut -c1-300
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-b1 --no-variant-b --sweep-steps 0 > $GRAFT_REPO_ROOT/$O/stats.log 2>&1)
ls -t $O/stats/*/*kernel_stats.csv | head -1 | xargs cat | head -5
(timeout 200 python tools/gpu_profile.py industrial_poly 16384; timeout 100 python tools/gpu_profile.py industrial_poly 256; timeout 100 python tools/gpu_profile.py industrial_poly 1) > $O/phase.txt 2>&1
(DOMPC_SCHED=default DOMPC_DEFS="DOMPC_OLDSCHED=1" timeout 200 python tools/gpu_profile.py industrial_poly 16384) > $O/phase_default_scheduler.txt 2>&1
grep -v "^/opt" $O/phase.txt | head -40
grep -v "^/opt" $O/phase_default_scheduler.txt | head -40
DOMPC_PMC_BATCH=16384 DOMPC_PMC_DIR=$R/pmc DOMPC_PMC_TIMEOUT=120 timeout 600 bash tools/pmc_run2.sh > $O/pmc.log 2>&1
tail -34 $O/pmc.log
timeout 300 bash tools/pmc_sweep.sh > $O/pmc_sweep.log 2>&1
grep "slots     0" $O/pmc_sweep.log | head -30
timeout 200 python bench.py --variant tree --steps 3 --warmup 1 2>/dev/null | grep '^{"metric' > $O/bench_tree.json; cut -c1-200 $O/bench_tree.json
