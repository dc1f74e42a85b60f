"""TEST-ONLY host emulation of the device code (one workgroup = one host thread).

The kernel sources are compiled with g++ -DDOMPC_HOST_EMU into tests/_hostemu/ so that the
IPM logic can be checked in the GPU-less CI container.  The product never loads these
libraries: do_mpc_amd.solver.HipIpmSolver defaults to the HIP runtime and raises without a GPU.
"""
import hostemu_build
import contextlib
import os

from do_mpc_amd import build, controller
from do_mpc_amd.solver import HipIpmSolver

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_hostemu")


def factory(structure, header, model_hash, nlpsol_opts=None, device=0, max_batch=1, block_threads=0, **kw):
    lib = hostemu_build.hostemu_library(header, model_hash, OUT)
    return HipIpmSolver(structure, header, model_hash, nlpsol_opts=nlpsol_opts, device=device, max_batch=max_batch,
                        _lib_path=lib, _code_object="")


@contextlib.contextmanager
def patched():
    orig = controller.HipIpmSolver
    controller.HipIpmSolver = factory
    try:
        yield
    finally:
        controller.HipIpmSolver = orig
