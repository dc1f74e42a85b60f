# Host emulation of the kernels under AddressSanitizer (the kernel text compiled by g++ with -fsanitize=address: LDS regions, records and
# workspaces are plain host arrays there, so out-of-bounds indexing of the kernels' own data shows up).  Serial (xdist workers do not start
# under the preloaded runtime); about 15 minutes.   bash tools/asan_hostemu.sh [pytest arguments]
cd "$(dirname "$0")/.."
ASAN=$(gcc -print-file-name=libasan.so)
DOMPC_DEFS="-fsanitize=address -fno-omit-frame-pointer -g" LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
  python -m pytest tests/test_hostemu_parity.py tests/test_edge_cases.py tests/test_differentiator.py tests/test_mhe.py tests/test_simulator.py -x -q "$@"
