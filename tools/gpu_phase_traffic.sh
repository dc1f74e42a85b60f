#!/bin/bash
# Time and HBM traffic of the phases of the batched solve, by difference: builds with -DDOMPC_REPEAT_PHASE=<mask> run the named phase twice
# per call (same iterates, same iteration counts), the bench line of each build carries kernel time and live PMC traffic of one launch.
#   python tools/ab_prebuild_plain.py "" DOMPC_REPEAT_PHASE=1 ... ; gpurun -- 'bash tools/gpu_phase_traffic.sh'
mkdir -p gpurun_out
O=gpurun_out/phase_traffic.txt
: > $O
one() {
  if [ -n "$1" ]; then export DOMPC_DEFS="$1"; else unset DOMPC_DEFS; fi
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variant-b --no-b1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'defs':'$1','kernel_ms':r['kernel_ms'],'traffic':r['traffic'],'iters':d['solve']['iters_mean'],'sweeps':d['solve']['sweeps_per_solve'],'trials':d['solve']['trials_per_solve'],'conv':d['solve']['converged'],'sweep_only_ms':r['sweep_only']['kernel_ms'],'sweep_only_traffic':r['sweep_only'].get('traffic')}))" | tee -a $O
}
echo -n "(warm-up, discarded) "; one ""
: > $O
for m in "" ${MASKS:-1 2 4 8 16} ""; do
  if [ -n "$m" ]; then one "DOMPC_REPEAT_PHASE=$m"; else one ""; fi
done
python - <<'PY' | tee -a gpurun_out/phase_traffic.txt
import json
rows=[json.loads(l) for l in open('gpurun_out/phase_traffic.txt') if l.startswith('{')]
base=[r for r in rows if not r['defs']]
b_ms=sum(r['kernel_ms'] for r in base)/len(base); b_tr=sum(r['traffic'] for r in base)/len(base)
B=16384
it=base[0]['iters']
print('base: %.1f ms, %.3f TB per launch (%d runs: %s ms)' % (b_ms, b_tr/1e12, len(base), ' / '.join('%.1f' % r['kernel_ms'] for r in base)))
d={}
for r in rows:
    if r['defs']:
        m=int(r['defs'].split('=')[1]); d[m]=(r['kernel_ms']-b_ms, r['traffic']-b_tr, r)
if 1 in d and 2 in d: d[2]=(d[2][0]-d[1][0], d[2][1]-d[1][1], d[2][2])
names={1:'sweep (model evaluation + edges + node assembly)',2:'backward Riccati pass',4:'forward pass (without the adjoint levels)',8:'trial evaluation',16:'step rules'}
acc_ms=acc_tr=0.0
for m in sorted(d):
    ms,tr,r=d[m]; acc_ms+=ms; acc_tr+=tr
    print('%-52s %7.1f ms %5.1f %%   %6.3f TB %5.1f %%   %6.0f doubles per edge and iteration   (iters %.3f conv %d)' % (names[m], ms, 100*ms/b_ms, tr/1e12, 100*tr/b_tr, tr/8/B/180/it, r['iters'], r['conv']))
print('%-52s %7.1f ms %5.1f %%   %6.3f TB %5.1f %%   %6.0f doubles per edge and iteration' % ('rest (measure, accept, driver, adjoint forward)', b_ms-acc_ms, 100*(b_ms-acc_ms)/b_ms, (b_tr-acc_tr)/1e12, 100*(b_tr-acc_tr)/b_tr, (b_tr-acc_tr)/8/B/180/it))
PY
