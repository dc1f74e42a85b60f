cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/tree
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tree/stats -- python $R/bench.py --variant tree --steps 2 --warmup 1 --cut-level 3 > $R/gpurun_out/tree/bench.log 2>&1
grep "^{" $R/gpurun_out/tree/bench.log | cut -c1-200
find $R/gpurun_out/tree/stats -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-200 | head -8
