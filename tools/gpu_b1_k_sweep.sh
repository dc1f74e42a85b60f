# one problem alone: workgroups per problem (K) of the whole-chip wide mode, cold make_step of industrial_poly / CSTR / batch_reactor N = 50
for k in 8 12 16 20 24 32 48; do DOMPC_WIDE=$k python tools/gpu_b1.py industrial_poly 2>&1 | grep "B=1 cold" | sed "s/^/K=$k /"; done
for k in 2 3 4 6 8; do DOMPC_WIDE=$k python tools/gpu_b1.py CSTR 2>&1 | grep "B=1 cold" | sed "s/^/K=$k /"; done
echo default; python tools/gpu_b1.py industrial_poly 2>&1 | grep "B=1 cold"; python tools/gpu_b1.py CSTR 2>&1 | grep "B=1 cold"
