"""Compile the native pieces: the generic runtime (C ABI) and per-model gfx950 code objects.

Everything is built in-tree under do_mpc_amd/_build/ (git-ignored; travels to the GPU box with
the gpurun snapshot).  hipcc cross-compiles for gfx950 without a GPU.
"""
from __future__ import annotations

import contextlib
import fcntl
import hashlib
import os
import shutil
import subprocess
import tempfile
from typing import Optional

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
BUILD = os.path.join(HERE, "_build")
ARCH = "gfx950"


class BuildError(RuntimeError):
    pass


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise BuildError("hipcc not found: the dompc IPM backend needs the ROCm toolchain to lower models to gfx950")


def _run(cmd, what):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise BuildError(f"{what} failed:\n{' '.join(cmd)}\n{r.stdout}")
    return r.stdout


@contextlib.contextmanager
def _locked(directory: str):
    """Exclusive lock on a build directory: several rank processes lower the same MPC at the same time
    (MPC.shard_tree, torchrun) and must not compile into the same paths concurrently."""
    os.makedirs(directory, exist_ok=True)
    fd = os.open(os.path.join(directory, ".lock"), os.O_CREAT | os.O_RDWR, 0o644)
    try:
        fcntl.flock(fd, fcntl.LOCK_EX)
        yield
    finally:
        fcntl.flock(fd, fcntl.LOCK_UN)
        os.close(fd)


def _write_atomic(path: str, text: str) -> None:
    """temp file in the same directory + rename: a reader never sees a truncated file"""
    fd, tmp = tempfile.mkstemp(dir=os.path.dirname(path), prefix=os.path.basename(path) + ".", suffix=".tmp")
    with os.fdopen(fd, "w") as f:
        f.write(text)
    os.replace(tmp, path)


def _fresh(out: str, stamp: str, dig: str) -> bool:
    try:
        return os.path.exists(out) and open(stamp).read() == dig
    except OSError:
        return False


def _compile_to(cmd_without_out, out: str, what: str) -> None:
    """run the compiler into a temporary name next to `out`, then rename into place"""
    tmp = f"{out}.{os.getpid()}.tmp"
    try:
        _run(cmd_without_out + ["-o", tmp], what)
        os.replace(tmp, out)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)


def _sources_digest() -> str:
    h = hashlib.sha256()
    for fn in sorted(os.listdir(CSRC)):                       # (every source of the kernels and of the runtime)
        if not fn.endswith((".h", ".hip", ".cpp")):
            continue
        h.update(fn.encode())
        with open(os.path.join(CSRC, fn), "rb") as f:
            h.update(f.read())
    with open(os.path.join(INCLUDE, "dompc_ipm.h"), "rb") as f:
        h.update(f.read())
    return h.hexdigest()[:12]


def runtime_library(force: bool = False) -> str:
    """libdompc_ipm.so - the C-ABI host runtime (HIP only; no CPU fallback inside)."""
    os.makedirs(BUILD, exist_ok=True)
    out = os.path.join(BUILD, "libdompc_ipm.so")
    stamp = out + ".stamp"
    dig = _sources_digest()
    if not force and _fresh(out, stamp, dig):
        return out
    with _locked(BUILD):
        if not force and _fresh(out, stamp, dig):       # another process built it while we waited
            return out
        cmd = [_hipcc(), "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__",
               os.path.join(CSRC, "dompc_runtime.cpp"), os.path.join(CSRC, "dompc_plant_runtime.cpp"), "-ldl"]
        _compile_to(cmd, out, "building libdompc_ipm.so")
        _write_atomic(stamp, dig)
    return out


def model_dir(model_hash: str) -> str:
    d = os.path.join(BUILD, "models", model_hash)
    os.makedirs(d, exist_ok=True)
    return d


def model_code_object(header_text: str, model_hash: str, force: bool = False, opt: str = "-O3", shard: bool = False, batch_only: bool = False) -> str:
    """Per-model gfx950 code object (hsaco) from the generated header + the kernel sources.
    shard=True: the variant with tree-sharding support (ownership masks, cross-rank exchanges).
    batch_only=True: the sibling `..._batch.hsaco` compiled with -DDOMPC_NO_WIDE=1 -DDOMPC_BLOCK_CONST=64 - "one workgroup of ONE wavefront
    per problem" (the launch shape of batches >= 4096) is a compile-time fact there: no device-scope barrier / atomic flag code in the phases
    (+1.6 % on the headline batch in a same-box A/B), thread counts and strides constants, workgroup barriers folded (+1.0 % more); the
    runtime loads it next to the general object when it exists and launches it whenever the workgroups have 64 threads."""
    forced = os.environ.get("DOMPC_CODE_OBJECT")     # measurement aid: use this code object as it is (A/B against an older kernel)
    if forced and not shard:
        return forced
    assert not (shard and batch_only)
    d = model_dir(model_hash)
    hdr = os.path.join(d, "model_gen.h")
    out = os.path.join(d, f"dompc_{ARCH}{'_shard' if shard else ''}.hsaco")
    stamp = out + ".stamp"
    lb = os.environ.get("DOMPC_LB", "2")          # tuning aid: wavefronts per SIMD the kernel is compiled for
    prof = os.environ.get("DOMPC_PROFILE", "0")    # measurement aid: sub-phase cycle counters compiled in (tools/gpu_profile.py)
    defs = os.environ.get("DOMPC_DEFS", "").split()  # measurement aid: extra -D switches / compiler flags (entries starting with '-') for A/B builds (own file per set)
    if defs or prof != "0" or lb != "2":          # (measurement builds live next to the product build, under their own names)
        tag = ("prof" if prof != "0" else "") + (("lb" + lb) if lb != "2" else "") + (hashlib.sha256(" ".join(defs).encode()).hexdigest()[:8] if defs else "")
        out = out[:-len(".hsaco")] + "_" + tag + ".hsaco"
    if batch_only:
        out = out[:-len(".hsaco")] + "_batch.hsaco"      # (the name the runtime derives from the general object's path)
        defs = defs + ["DOMPC_NO_WIDE=1", "DOMPC_BLOCK_CONST=64"]
    stamp = out + ".stamp"
    # machine scheduler / register allocation of the solver kernels (round 5, same-box A/B on the headline batch, profiles/r05_flags.txt):
    # the iterative ILP strategy of the AMDGPU backend +3.0 % against the default strategy, region priorities in the greedy allocator
    # another +0.6 % (default 7 604 -> 7 828 / 7 853 -> 7 889 MPC steps/s; iterative-minreg -9 %, iterative-maxocc +2.5 %, post-RA scheduler
    # off -1.8 %; bitwise the same results: the flags reorder instructions, they do not change the arithmetic).  DOMPC_SCHED=default
    # builds without them; a DOMPC_DEFS set that names a strategy itself wins.
    sched = []
    if os.environ.get("DOMPC_SCHED", "iterative-ilp") != "default" and not any("amdgpu-sched-strategy" in d_ for d_ in defs):
        sched = ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]
        if not any("greedy-regclass-priority" in d_ for d_ in defs):
            sched += ["-mllvm", "-greedy-regclass-priority-trumps-globalness=1"]
    dig = _sources_digest() + hashlib.sha256(header_text.encode()).hexdigest()[:12] + opt + ("S" if shard else "") + "lb" + lb + "p" + prof + " ".join(defs) + " ".join(sched)
    if not force and _fresh(out, stamp, dig):
        return out
    with _locked(d):
        if not force and _fresh(out, stamp, dig):       # another rank built it while we waited for the lock
            return out
        if not (os.path.exists(hdr) and open(hdr).read() == header_text):
            _write_atomic(hdr, header_text)
        # (the iterative scheduler of this ROCm's clang crashes - segmentation fault in the greedy register allocator - on some models, e.g. the
        #  discrete oscillating-masses class: such a model is compiled with the default strategy instead; `<out>.sched` says which one it got)
        last = None
        for fl in ([sched, []] if sched else [[]]):
            cmd = [_hipcc(), f"--offload-arch={ARCH}", opt, "-std=c++17", "--genco", f"-DDOMPC_SHARD={1 if shard else 0}", f"-DDOMPC_SRC_DIGEST=0x{_sources_digest()}ULL", *fl, f"-DDOMPC_LB={lb}", f"-DDOMPC_PROFILE={prof}", *[(d if d.startswith("-") else f"-D{d}") for d in defs],
                   f"-DDOMPC_MODEL_HEADER=\"{hdr}\"", "-I", CSRC, os.path.join(CSRC, "dompc_device.hip")]
            try:
                _compile_to(cmd, out, f"lowering model {model_hash} to {ARCH}")
                _write_atomic(out + ".sched", " ".join(fl) or "default")
                last = None
                break
            except BuildError as e:
                last = e
                if "Segmentation fault" not in str(e) and "frontend command failed" not in str(e):
                    raise
        if last is not None:
            raise last
        _write_atomic(stamp, dig)
    return out


def plant_code_object(header_text: str, model_hash: str, force: bool = False) -> str:
    """gfx950 code object of the batched plant integrator (csrc/dompc_plant.hip) for one lowered plant model."""
    d = model_dir("plant_" + model_hash)
    hdr = os.path.join(d, "plant_gen.h")
    out = os.path.join(d, f"dompc_plant_{ARCH}.hsaco")
    stamp = out + ".stamp"
    dig = _sources_digest() + hashlib.sha256(header_text.encode()).hexdigest()[:12]
    if not force and _fresh(out, stamp, dig):
        return out
    with _locked(d):
        if not force and _fresh(out, stamp, dig):
            return out
        if not (os.path.exists(hdr) and open(hdr).read() == header_text):
            _write_atomic(hdr, header_text)
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "--genco", f"-DDOMPC_PLANT_HEADER=\"{hdr}\"", "-I", CSRC,
               os.path.join(CSRC, "dompc_plant.hip")]
        _compile_to(cmd, out, f"lowering plant {model_hash} to {ARCH}")
        _write_atomic(stamp, dig)
    return out
