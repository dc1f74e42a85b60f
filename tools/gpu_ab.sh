#!/bin/bash
# same-box A/B of build switches on the headline batch: tools/gpu_ab.sh "<defs A>" "<defs B>" ... (an empty string = the product build);
# every variant is run twice, interleaved; one line per run: value, ms per launch, sweep-only ms, converged, mean iterations
mkdir -p gpurun_out
for rep in 1 2; do
  for defs in "$@"; do
    if [ -n "$defs" ]; then export DOMPC_DEFS="$defs"; else unset DOMPC_DEFS; fi
    python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-variant-b --no-b1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('[%s] %.1f steps/s  %.1f ms  sweep-only %.2f ms  conv %d  iters %.3f' % ('$defs', d['value'], d['ms_per_step'], d['roofline']['sweep_only']['kernel_ms'], d['solve']['converged'], d['solve']['iters_mean']))"
  done
done
