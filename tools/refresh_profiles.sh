cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01b
python bench.py --steps 3 --warmup 1 > gpurun_out/r01b/bench.json 2> gpurun_out/r01b/bench.err
cut -c1-300 gpurun_out/r01b/bench.json
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r01b/stats -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r01b/stats.log 2>&1)
ls -t gpurun_out/r01b/stats/*/*kernel_stats.csv | head -1 | xargs cat | head -5
(python tools/gpu_profile.py industrial_poly 1024; python tools/gpu_profile.py industrial_poly 1; python tools/gpu_check.py 2>&1 | tail -12) > gpurun_out/r01b/phase.txt 2>&1
bash tools/pmc_run.sh > gpurun_out/r01b/pmc.log 2>&1
tail -3 gpurun_out/r01b/pmc.log
