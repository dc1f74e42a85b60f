# A/B of DOMPC_DEFS sets in ONE GPU call (same box): bench (3 steps) + phase profile per set.
#   bash tools/ab_run.sh tag "DEFS1" "DEFS2" ...
cd $GRAFT_REPO_ROOT
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
i=0
for D in "$@"; do
  i=$((i+1))
  echo "== [$i] DOMPC_DEFS='$D'" | tee -a $O/ab.txt
  export DOMPC_LB=2
  case "$D" in LB=*) export DOMPC_LB=${D:3:1}; D="${D:5}";; esac
  DOMPC_DEFS="$D" timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-b1 2> $O/b$i.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   steps/s %.1f  kernel_ms %.2f  converged %d  iters %.3f u0 %r' % (d['value'], d['roofline']['kernel_ms'], d['solve']['converged'], d['solve']['iters_mean'], d['solve']['u0_first']))" | tee -a $O/ab.txt
  DOMPC_DEFS="$D" timeout 200 python tools/gpu_profile.py industrial_poly 4096 2>/dev/null | grep -v "^/opt" | tee -a $O/ab.txt
done
