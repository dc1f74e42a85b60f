# A/B of DOMPC_DEFS sets against the product build and a saved base code object, interleaved, same box:
#   bash tools/ab_defs.sh tag batch "DEFS1" ...      ('' = product build; BASE = gpurun_ab/ip_base.hsaco)
cd $GRAFT_REPO_ROOT
TAG=$1; B=$2; shift; shift
O=gpurun_out/$TAG; mkdir -p $O
run() {
  python bench.py --steps 3 --warmup 1 --batch $B --no-cpu-baseline --no-traffic --no-b1 2> $O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   steps/s %.1f  kernel_ms %.2f  converged %d  iters %.3f u0 %r' % (d['value'], d['roofline']['kernel_ms'], d['solve']['converged'], d['solve']['iters_mean'], d['solve']['u0_first']))"
}
for rep in 1 2; do
  for D in "$@"; do
    echo "== [$D]" | tee -a $O/ab.txt
    if [ "$D" = "BASE" ]; then DOMPC_CODE_OBJECT=$GRAFT_REPO_ROOT/gpurun_ab/ip_base.hsaco run | tee -a $O/ab.txt
    else DOMPC_DEFS="$D" run | tee -a $O/ab.txt; fi
  done
done
