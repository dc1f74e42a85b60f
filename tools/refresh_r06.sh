# The measured artefacts of round 6 in one GPU call: smoke, the driver's bench command, rocprofv3 kernel statistics of the same launches,
# phase counters (problem 0's wavefront clock), the PMC passes of the solve and of the sweep-only kernel, the same-box A/B of the quad
# sweep against the wavefront-per-edge sweep it replaces, the randomly drawn members of the timed batch, the 243-leaf tree.
#   bash tools/refresh_r06.sh [tag]   -> gpurun_out/<tag>/
R=${1:-r06g}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$R
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
grep '^{"metric' $O/bench.json | cut -c1-300
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-b1 --no-variant-b --sweep-steps 0 > $GRAFT_REPO_ROOT/$O/stats.log 2>&1)
ls -t $O/stats/*/*kernel_stats.csv | head -1 | xargs cat | head -5
(timeout 200 python tools/gpu_profile.py industrial_poly 16384; timeout 100 python tools/gpu_profile.py industrial_poly 256; timeout 100 python tools/gpu_profile.py industrial_poly 1) > $O/phase.txt 2>&1
grep -v "^/opt" $O/phase.txt | head -40
tools/gpu_ab.sh "" "DOMPC_QUAD=0" > $O/ab_quad.txt 2>&1; cat $O/ab_quad.txt
DOMPC_PMC_BATCH=16384 DOMPC_PMC_DIR=$R/pmc DOMPC_PMC_TIMEOUT=120 timeout 600 bash tools/pmc_run2.sh > $O/pmc.log 2>&1
tail -34 $O/pmc.log
timeout 400 bash tools/pmc_sweep.sh > $O/pmc_sweep.log 2>&1
cp gpurun_out/pmc_sweep/summary_slots0.json $O/sweep_counters_slots0.json 2>/dev/null; cp gpurun_out/pmc_sweep/summary_slots1024.json $O/sweep_counters_slots1024.json 2>/dev/null
grep "slots     0" $O/pmc_sweep.log | head -30
timeout 400 python tools/gpu_random_members.py 64 2026 > $O/random_members.txt 2>&1; tail -3 $O/random_members.txt
timeout 200 python bench.py --variant tree --steps 3 --warmup 1 2>/dev/null | grep '^{"metric' > $O/bench_tree.json; cut -c1-200 $O/bench_tree.json
timeout 300 python bench.py --variant closed_loop --steps 3 --warmup 1 2>/dev/null | grep '^{"metric' > $O/bench_closed_loop.json; cut -c1-200 $O/bench_closed_loop.json
