#!/bin/bash
# same-box A/B of build switches on the headline batch: tools/gpu_ab.sh "<defs A>" "<defs B>" ... (an empty string = the product build).
# One discarded warm-up run (the first run on a fresh box is 3 - 6 % faster than every later one: clocks), then every variant REPS times,
# interleaved; one line per run: value, ms per launch, sweep-only ms, converged, mean iterations
mkdir -p gpurun_out
REPS=${REPS:-2}
one() {
  if [ -n "$1" ]; then export DOMPC_DEFS="$1"; else unset DOMPC_DEFS; fi
  python bench.py --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline --no-traffic --no-variant-b --no-b1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('[%s] %.1f steps/s  %.1f ms  sweep-only %.2f ms  conv %d  iters %.3f' % ('$1', d['value'], d['ms_per_step'], d['roofline']['sweep_only']['kernel_ms'], d['solve']['converged'], d['solve']['iters_mean']))"
}
echo -n "(warm-up, discarded) "; one "$1"
for rep in $(seq $REPS); do
  for defs in "$@"; do one "$defs"; done
done
