# A/B of the current tree against a saved code object (gpurun_ab/ip_base.hsaco: the product build of an earlier commit), same box:
#   bash tools/ab_base.sh tag [batch]
cd $GRAFT_REPO_ROOT
TAG=$1; B=${2:-4096}
O=gpurun_out/$TAG; mkdir -p $O
run() {
  python bench.py --steps 3 --warmup 1 --batch $B --no-cpu-baseline --no-traffic --no-b1 2> $O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   steps/s %.1f  kernel_ms %.2f  converged %d  iters %.3f u0 %r' % (d['value'], d['roofline']['kernel_ms'], d['solve']['converged'], d['solve']['iters_mean'], d['solve']['u0_first']))"
}
for rep in 1 2; do
  echo "== base" | tee -a $O/ab.txt;  DOMPC_CODE_OBJECT=$GRAFT_REPO_ROOT/gpurun_ab/ip_base.hsaco run | tee -a $O/ab.txt
  echo "== new" | tee -a $O/ab.txt;   run | tee -a $O/ab.txt
done
DOMPC_PROFILE=1 timeout 200 python tools/gpu_profile.py industrial_poly $B 2>/dev/null | grep -v "^/opt" | tee -a $O/ab.txt
