"""TEST-ONLY builders of the host emulation: the kernel sources compiled by g++ with -DDOMPC_HOST_EMU (one "workgroup" = one host
thread) into tests/_hostemu/.  They lived in do_mpc_amd/build.py until round 4 although only tests call them: the product
package now holds nothing that can run the solver without a GPU."""
import hashlib
import os
import shutil

from do_mpc_amd.build import CSRC, _compile_to, _fresh, _locked, _sources_digest, _write_atomic


def hostemu_library(header_text: str, model_hash: str, out_dir: str, force: bool = False) -> str:
    """TEST-ONLY: runtime + kernels compiled for the host (g++), one workgroup = one thread.
    Lives outside the package build dir (tests/_hostemu) and is never loaded by the product."""
    os.makedirs(out_dir, exist_ok=True)
    hdr = os.path.join(out_dir, f"model_gen_{model_hash}.h")
    defs = os.environ.get("DOMPC_DEFS", "").split()      # extra -D switches (e.g. DOMPC_KAPPA_D=1e-5): an own library per set
    tag = ("_" + hashlib.sha256(" ".join(defs).encode()).hexdigest()[:8]) if defs else ""
    out = os.path.join(out_dir, f"libdompc_hostemu_{model_hash}{tag}.so")
    stamp = out + ".stamp"
    dig = _sources_digest() + hashlib.sha256(header_text.encode()).hexdigest()[:12] + " ".join(defs)
    if not force and _fresh(out, stamp, dig):
        return out
    with _locked(out_dir):                                # (world_size-2 gloo tests build from two processes)
        if not force and _fresh(out, stamp, dig):
            return out
        if not (os.path.exists(hdr) and open(hdr).read() == header_text):
            _write_atomic(hdr, header_text)
        cxx = shutil.which("g++") or "g++"
        cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-DDOMPC_HOST_EMU", "-DDOMPC_SHARD=1",
               *[(d if d.startswith("-") else f"-D{d}") for d in defs],
               f"-DDOMPC_MODEL_HEADER=\"{hdr}\"", "-I", CSRC,
               os.path.join(CSRC, "dompc_runtime.cpp"), "-x", "c++", os.path.join(CSRC, "dompc_device.hip"), "-lm"]
        _compile_to(cmd, out, "building host emulation")
        _write_atomic(stamp, dig)
    return out




def plant_hostemu_library(header_text: str, model_hash: str, out_dir: str, force: bool = False) -> str:
    """TEST-ONLY: plant integrator compiled for the host (g++); lives in tests/_hostemu, never loaded by the product."""
    os.makedirs(out_dir, exist_ok=True)
    hdr = os.path.join(out_dir, f"plant_gen_{model_hash}.h")
    out = os.path.join(out_dir, f"libdompc_plant_hostemu_{model_hash}.so")
    stamp = out + ".stamp"
    dig = _sources_digest() + hashlib.sha256(header_text.encode()).hexdigest()[:12]
    if not force and _fresh(out, stamp, dig):
        return out
    with _locked(out_dir):
        if not force and _fresh(out, stamp, dig):
            return out
        if not (os.path.exists(hdr) and open(hdr).read() == header_text):
            _write_atomic(hdr, header_text)
        cxx = shutil.which("g++") or "g++"
        cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-DDOMPC_HOST_EMU", f"-DDOMPC_PLANT_HEADER=\"{hdr}\"", "-I", CSRC,
               os.path.join(CSRC, "dompc_plant_runtime.cpp"), "-x", "c++", os.path.join(CSRC, "dompc_plant.hip"), "-lm"]
        _compile_to(cmd, out, "building plant host emulation")
        _write_atomic(stamp, dig)
    return out
