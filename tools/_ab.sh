for rep in 1 2 3; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-variant-b --sweep-steps 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('with batch object   ', round(d['value'],1), d['solve']['converged'], 'b1', round(d['make_step_ms_b1']['cold'],2))"
  DOMPC_NO_BATCH_OBJECT=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-variant-b --sweep-steps 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('general object only ', round(d['value'],1), d['solve']['converged'], 'b1', round(d['make_step_ms_b1']['cold'],2))"
done
