# 243-leaf tree (BASELINE configs[4]), one problem alone: workgroups per problem of the whole-chip wide mode
for k in 32 48 64 80 96 112 128; do DOMPC_WIDE=$k python bench.py --variant tree --steps 5 --warmup 2 2>/dev/null | grep '"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('K=$k  ms_per_step', round(d['ms_per_step'], 2), {k: v for k, v in d.items() if k in ('iters', 'converged', 'workgroups_per_problem')}, {k: v for k, v in d.get('config', {}).items() if 'work' in k or 'K' == k})"; done
