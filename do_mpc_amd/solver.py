"""ctypes binding of the C ABI (include/dompc_ipm.h) - the object stored in MPC.S.

Drop-in for what `castools.nlpsol('S','ipopt',nlp,opts)` returns at
/root/reference/do_mpc/controller/_mpc.py:1328: callable with keyword arguments
x0, lbx, ubx, lbg, ubg, p (lam_x0, lam_g0 accepted and ignored, as IPOPT does with
warm_start_init_point=no) returning {'x','f','g','lam_x','lam_g'}, plus .stats()
(/root/reference/do_mpc/optimizer.py:754-778).

There is no CPU fallback: constructing the solver without the HIP runtime / a GPU raises.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import time
from typing import Dict, Optional

import numpy as np

from . import build
from .structure import ProblemStructure

_i32p = C.POINTER(C.c_int32)
_f64p = C.POINTER(C.c_double)


class Options(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "tol", "dual_inf_tol", "constr_viol_tol", "compl_inf_tol", "acceptable_tol", "mu_init", "kappa_mu",
        "theta_mu", "kappa_eps", "tau_min", "bound_push", "bound_frac", "bound_relax_factor",
        "nlp_scaling_max_gradient", "delta_w_0", "delta_w_min", "delta_w_max", "kappa_w_minus", "kappa_w_plus",
        "kappa_w_plus_bar")] + [(n, C.c_int32) for n in ("max_iter", "acceptable_iter", "obj_scaling", "max_soc")] + [
        ("constr_mult_init_max", C.c_double), ("watchdog_shortened_iter_trigger", C.c_int32), ("watchdog_trial_iter_max", C.c_int32)]


class ProblemDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "nx", "nu", "np", "ntvp", "ne", "ns", "deg", "ni", "M", "N", "n_opt_x", "n_opt_p", "n_g", "n_nodes",
        "n_edges", "n_dummy", "p_off_tvp", "p_off_p", "p_off_uprev")] + \
        [(n, _i32p) for n in (
            "level_node_start", "node_level", "node_x_off", "node_u_off", "node_eps_off", "node_child_start",
            "node_child_count", "node_parent", "node_in_edge", "edge_parent", "edge_child", "edge_pidx",
            "edge_w_off", "edge_row0", "edge_level")] + \
        [("edge_omega", _f64p), ("dummy_idx", _i32p), ("code_object_path", C.c_char_p), ("model_hash", C.c_char_p),
         ("device", C.c_int32), ("max_batch", C.c_int32), ("n_slots", C.c_int32), ("block_threads", C.c_int32),
         ("opts", Options)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("success", "status", "iter_count", "n_reg", "n_ls_fail", "n_sweeps", "n_trials", "n_soc", "n_watchdog",
                                             "reserved0")] + \
               [(n, C.c_double) for n in ("mu", "obj", "inf_pr", "inf_du", "inf_compl", "obj_scaling", "t_wall_total")]


_ALLREDUCE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int32)


class ShardDesc(C.Structure):
    """dompc_shard_desc (include/dompc_ipm.h)"""
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("cut_level", C.c_int32), ("n_cut", C.c_int32),
                ("x_mask", C.c_void_p), ("g_mask", C.c_void_p), ("edge_mask", C.c_void_p), ("node_mask", C.c_void_p),
                ("node_cut", C.c_void_p), ("xbuf", C.c_void_p), ("xbuf_doubles", C.c_int64), ("allreduce", _ALLREDUCE_FN),
                ("ctx", C.c_void_p)]


STATS_DTYPE = np.dtype([("success", "i4"), ("status", "i4"), ("iter_count", "i4"), ("n_reg", "i4"),
                        ("n_ls_fail", "i4"), ("n_sweeps", "i4"), ("n_trials", "i4"), ("n_soc", "i4"), ("n_watchdog", "i4"), ("reserved0", "i4"), ("mu", "f8"), ("obj", "f8"), ("inf_pr", "f8"),
                        ("inf_du", "f8"), ("inf_compl", "f8"), ("obj_scaling", "f8"), ("t_wall_total", "f8")])
assert STATS_DTYPE.itemsize == C.sizeof(Stats)

# ipopt option name -> field
_IPOPT_OPTS = {
    "ipopt.tol": "tol", "ipopt.dual_inf_tol": "dual_inf_tol", "ipopt.constr_viol_tol": "constr_viol_tol",
    "ipopt.compl_inf_tol": "compl_inf_tol", "ipopt.acceptable_tol": "acceptable_tol", "ipopt.mu_init": "mu_init",
    "ipopt.mu_linear_decrease_factor": "kappa_mu", "ipopt.mu_superlinear_decrease_power": "theta_mu",
    "ipopt.barrier_tol_factor": "kappa_eps", "ipopt.tau_min": "tau_min", "ipopt.bound_push": "bound_push",
    "ipopt.bound_frac": "bound_frac", "ipopt.bound_relax_factor": "bound_relax_factor",
    "ipopt.nlp_scaling_max_gradient": "nlp_scaling_max_gradient", "ipopt.max_iter": "max_iter",
    "ipopt.acceptable_iter": "acceptable_iter",
    "ipopt.first_hessian_perturbation": "delta_w_0", "ipopt.min_hessian_perturbation": "delta_w_min",
    "ipopt.max_hessian_perturbation": "delta_w_max", "ipopt.max_soc": "max_soc",
    "ipopt.constr_mult_init_max": "constr_mult_init_max",
    "ipopt.watchdog_shortened_iter_trigger": "watchdog_shortened_iter_trigger", "ipopt.watchdog_trial_iter_max": "watchdog_trial_iter_max",
}


def _load(lib_path: str) -> C.CDLL:
    lib = C.CDLL(lib_path)
    lib.dompc_default_options.argtypes = [C.POINTER(Options)]
    lib.dompc_create.argtypes = [C.POINTER(ProblemDesc), C.POINTER(C.c_void_p)]
    lib.dompc_create.restype = C.c_int
    lib.dompc_destroy.argtypes = [C.c_void_p]
    lib.dompc_last_error.argtypes = [C.c_void_p]
    lib.dompc_last_error.restype = C.c_char_p
    lib.dompc_status_string.argtypes = [C.c_int32]
    lib.dompc_status_string.restype = C.c_char_p
    vp = C.c_void_p
    lib.dompc_solve.argtypes = [vp] + [vp] * 8 + [vp] * 5 + [vp]
    lib.dompc_solve.restype = C.c_int
    lib.dompc_solve_batch.argtypes = [vp, C.c_int32] + [vp] * 6 + [vp] * 5 + [vp]
    lib.dompc_solve_batch.restype = C.c_int
    lib.dompc_solve_batch_device.argtypes = [vp, C.c_int32] + [vp] * 6 + [vp] * 5 + [vp, vp]
    lib.dompc_solve_batch_device.restype = C.c_int
    lib.dompc_sweep_batch_device.argtypes = [vp, C.c_int32] + [vp] * 5 + [vp]
    lib.dompc_sweep_batch_device.restype = C.c_int
    lib.dompc_sweep_block_doubles.argtypes = [vp]
    lib.dompc_sweep_block_doubles.restype = C.c_int64
    lib.dompc_debug_newton_step.argtypes = [vp] + [vp] * 9 + [C.c_double, C.c_double] + [vp] * 4
    lib.dompc_debug_newton_step.restype = C.c_int
    lib.dompc_newton_step_at_solution.argtypes = [vp] + [vp] * 9 + [C.c_double] + [vp] * 2
    lib.dompc_newton_step_at_solution.restype = C.c_int
    lib.dompc_newton_steps_at_solution.argtypes = [vp, C.c_int32] + [vp] * 9 + [C.c_double] + [vp] * 2
    lib.dompc_newton_steps_at_solution.restype = C.c_int
    lib.dompc_debug_get_trace.argtypes = [vp, vp, C.c_int32]
    lib.dompc_debug_get_trace.restype = C.c_int
    lib.dompc_abort.argtypes = [vp, C.c_int32]
    lib.dompc_abort.restype = C.c_int
    lib.dompc_workspace_bytes.argtypes = [vp]
    lib.dompc_workspace_bytes.restype = C.c_int64
    lib.dompc_num_slots.argtypes = [vp]
    lib.dompc_num_slots.restype = C.c_int32
    lib.dompc_exchange_doubles.argtypes = [vp, C.c_int32, C.c_int32]
    lib.dompc_exchange_doubles.restype = C.c_int64
    lib.dompc_set_sharding.argtypes = [vp, C.POINTER(ShardDesc)]
    lib.dompc_set_sharding.restype = C.c_int
    lib.dompc_last_exchange_count.argtypes = [vp]
    lib.dompc_last_exchange_count.restype = C.c_int64
    lib.dompc_batch_object_state.argtypes = [vp]
    lib.dompc_batch_object_state.restype = C.c_int
    lib.dompc_edges_per_wavefront.argtypes = [vp]
    lib.dompc_edges_per_wavefront.restype = C.c_int
    lib.dompc_rccl_unique_id.argtypes = [vp, C.c_char_p, vp]
    lib.dompc_rccl_unique_id.restype = C.c_int
    lib.dompc_rccl_init.argtypes = [vp, C.c_char_p, vp, C.c_int32, C.c_int32]
    lib.dompc_rccl_init.restype = C.c_int
    return lib


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a, n=None) -> np.ndarray:
    if hasattr(a, "master"):
        a = a.master
    elif hasattr(a, "arr"):
        a = a.arr
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))
    if n is not None and a.size != n:
        raise ValueError(f"expected {n} values, got {a.size}")
    return a


class HipIpmSolver:
    """One handle = one problem class on one GPU."""

    def __init__(self, structure: ProblemStructure, header_text: str, model_hash: str,
                 nlpsol_opts: Optional[dict] = None, device: int = 0, max_batch: int = 1, n_slots: int = 0,
                 block_threads: int = 0, shard: bool = False, _lib_path: Optional[str] = None,
                 _code_object: Optional[str] = None):
        self.structure = ps = structure
        self.model_hash = model_hash
        self._ctor = dict(structure=structure, header_text=header_text, model_hash=model_hash, nlpsol_opts=nlpsol_opts,
                          device=device, max_batch=max_batch, n_slots=n_slots, block_threads=block_threads)
        self.shard_capable = bool(shard) or _code_object == ""
        if _lib_path is None:
            import os
            if not os.environ.get("DOMPC_NO_TORCH_FIRST"):
                try:                      # torch ships its own HIP runtime: it has to be the first one in the process
                    import torch          # noqa: F401
                    torch.cuda.is_available()
                except ImportError:
                    pass
            _lib_path = build.runtime_library()
            _code_object = build.model_code_object(header_text, model_hash, shard=bool(shard))
            sibling = _code_object[:-len(".hsaco")] + "_batch.hsaco"
            # (a sibling left behind by an earlier handle is refreshed with the general object: the runtime launches whatever lies there)
            if not shard and (int(max_batch) >= 4096 or int(block_threads) == 64 or os.path.exists(sibling)) and not os.environ.get("DOMPC_CODE_OBJECT"):
                # handles that solve large batches (one wavefront per problem from B = 4096 on) also get the build of the kernels that is
                # compiled for exactly that launch shape: the runtime finds it next to the general object
                build.model_code_object(header_text, model_hash, batch_only=True)
        self._lib = _load(_lib_path)
        self._keep = []
        d = ProblemDesc()
        for k, v in dict(nx=ps.nx, nu=ps.nu, np=ps.np_, ntvp=ps.ntvp, ne=ps.ne, ns=ps.ns, deg=ps.deg, ni=ps.ni,
                         M=ps.M, N=ps.N, n_opt_x=ps.n_opt_x, n_opt_p=ps.n_opt_p, n_g=ps.n_g, n_nodes=ps.n_nodes,
                         n_edges=ps.n_edges, n_dummy=len(ps.tables["dummy_idx"]), p_off_tvp=ps.p_off_tvp,
                         p_off_p=ps.p_off_p, p_off_uprev=ps.p_off_uprev).items():
            setattr(d, k, int(v))
        for name in ("level_node_start", "node_level", "node_x_off", "node_u_off", "node_eps_off",
                     "node_child_start", "node_child_count", "node_parent", "node_in_edge", "edge_parent",
                     "edge_child", "edge_pidx", "edge_w_off", "edge_row0", "edge_level", "dummy_idx"):
            arr = np.ascontiguousarray(ps.tables[name], dtype=np.int32)
            self._keep.append(arr)
            setattr(d, name, arr.ctypes.data_as(_i32p))
        om = np.ascontiguousarray(ps.tables["edge_omega"], dtype=np.float64)
        self._keep.append(om)
        d.edge_omega = om.ctypes.data_as(_f64p)
        d.code_object_path = (_code_object or "").encode()
        self.code_object_path = _code_object or ""          # the gfx950 code object of this problem class (named by the model hash)
        d.model_hash = model_hash.encode()
        d.device, d.max_batch, d.n_slots, d.block_threads = device, max_batch, n_slots, block_threads
        self._lib.dompc_default_options(C.byref(d.opts))
        self.ignored_options = []
        for k, v in (nlpsol_opts or {}).items():
            f = _IPOPT_OPTS.get(k)
            if f is not None:
                setattr(d.opts, f, type(getattr(d.opts, f))(v))
            elif k == "dompc.obj_scaling":
                d.opts.obj_scaling = int(v)
            elif k == "ipopt.kappa_d":
                # IPOPT's damping of one-sided bounds is a compile-time constant of the kernels (DOMPC_KAPPA_D = 1e-5, IPOPT's
                # default; -DDOMPC_KAPPA_D=... through DOMPC_DEFS builds another value): say so instead of ignoring it silently
                if float(v) != 1e-5:
                    import warnings
                    warnings.warn("ipopt.kappa_d = %r is not applied: the kernels are built with kappa_d = 1e-5 "
                                  "(DOMPC_DEFS='DOMPC_KAPPA_D=%r' builds a code object with this value)" % (v, v))
                self.ignored_options.append(k)
            else:
                self.ignored_options.append(k)     # print levels, linear solver, ... : no meaning here
        self.options = d.opts
        h = C.c_void_p()
        rc = self._lib.dompc_create(C.byref(d), C.byref(h))
        if rc != 0:
            raise RuntimeError("dompc_create failed: " + (self._lib.dompc_last_error(None) or b"?").decode())
        self._h = h
        # 0: no launch-shape-specific sibling code object, 1: loaded, 2: found but stale (other sources / model) and therefore not used
        self.batch_object_state = int(self._lib.dompc_batch_object_state(h))
        self.edges_per_wavefront = int(self._lib.dompc_edges_per_wavefront(h))      # 4: quad sweep (csrc/dompc_quad.h)
        if self.batch_object_state == 2:
            import warnings
            warnings.warn("dompc: the `_batch` code object next to %s was built from other sources and is not used "
                          "(one-wavefront batches run the general code object)" % (_code_object,))
        self._stats: Dict = {}
        self._device = device
        self._host_emulation = _code_object == ""
        self._shard = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.dompc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError("dompc: " + (self._lib.dompc_last_error(self._h) or b"?").decode())

    # ------------------------------------------------------------------ nlpsol-like call
    def __call__(self, x0, lbx, ubx, lbg, ubg, p, lam_x0=None, lam_g0=None) -> dict:
        ps = self.structure
        x0, lbx, ubx = _f64(x0, ps.n_opt_x), _f64(lbx, ps.n_opt_x), _f64(ubx, ps.n_opt_x)
        lbg, ubg, p = _f64(lbg, ps.n_g), _f64(ubg, ps.n_g), _f64(p, ps.n_opt_p)
        x = np.empty(ps.n_opt_x)
        g = np.empty(ps.n_g)
        lam_x = np.empty(ps.n_opt_x)
        lam_g = np.empty(ps.n_g)
        f = np.empty(1)
        st = np.zeros(1, dtype=STATS_DTYPE)
        self._check(self._lib.dompc_solve(self._h, _ptr(x0), _ptr(lbx), _ptr(ubx), _ptr(lbg), _ptr(ubg), _ptr(p),
                                          None, None, _ptr(x), _ptr(g), _ptr(lam_x), _ptr(lam_g), _ptr(f), _ptr(st)))
        self._stats = self._stats_dict(st[0])
        if self._shard is not None:
            self._stats["n_exchanges"] = int(self._lib.dompc_last_exchange_count(self._h))      # (host emulation: direct callbacks, 0)
            # every entry was written by exactly one rank (zeros elsewhere): the sum is the full vector
            x, g, lam_x, lam_g = (self._sum_over_ranks(a) for a in (x, g, lam_x, lam_g))
        return {"x": x, "f": float(f[0]), "g": g, "lam_x": lam_x, "lam_g": lam_g, "lam_p": np.zeros(ps.n_opt_p)}

    # ------------------------------------------------------------------ tree sharding over ranks
    def enable_sharding(self, rank: int, world: int, cut_level: Optional[int] = None, group=None, allreduce=None,
                        native_rccl: bool = True) -> dict:
        """Shard the scenario tree of this handle's problem over `world` ranks (one process per GPU).

        The sub-trees below the cut go to the ranks in contiguous blocks, the stages above are replicated; during a
        solve the kernel asks the host for element-wise SUMs over a small exchange buffer (cut-edge contributions
        of the Riccati recursion, scalar reductions of the IPM).  On the GPU the runtime joins its own RCCL
        communicator (unique id broadcast over `group`) and calls ncclAllReduce itself from the service loop
        (`native_rccl`); otherwise the collective is `torch.distributed.all_reduce` on `group` (nccl, or gloo in the
        CPU tests) or the callable `allreduce(view)`.  Every rank must call the solver with identical inputs."""
        from .structure import shard_tables
        if not self.shard_capable:
            raise RuntimeError("this solver was built without tree-sharding support: construct it with shard=True "
                               "(MPC.shard_tree does that)")
        t = shard_tables(self.structure, rank, world, cut_level)
        if t["cut_level"] < 1:
            raise ValueError("sharding needs a cut level >= 1 (a tree with n_robust >= 1)")
        n = int(self._lib.dompc_exchange_doubles(self._h, world, t["n_cut"]))
        import torch
        if self._host_emulation:
            xbuf = torch.zeros(n, dtype=torch.float64)
            stream = None
        else:
            xbuf = torch.zeros(n, dtype=torch.float64, device=torch.device("cuda", self._device))
            stream = torch.cuda.Stream(device=self._device)          # never the default stream: the solver kernel is resident
        base = xbuf.data_ptr()

        def reduce_view(view):
            if allreduce is not None:
                allreduce(view)
                return
            import torch.distributed as dist
            if stream is None:
                dist.all_reduce(view, group=group)
            else:
                with torch.cuda.stream(stream):
                    dist.all_reduce(view, group=group)
                stream.synchronize()

        def callback(_ctx, buf, count):
            off = (int(buf) - base) // 8
            reduce_view(xbuf[off:off + int(count)])

        native = bool(native_rccl) and allreduce is None and not self._host_emulation
        if native:
            import os
            import torch.distributed as dist
            path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so").encode()
            uid = torch.zeros(128, dtype=torch.uint8)
            if rank == 0:
                raw = (C.c_uint8 * 128)()
                self._check(self._lib.dompc_rccl_unique_id(self._h, path, raw))
                uid = torch.tensor(list(raw), dtype=torch.uint8)
            if world > 1:
                uid = uid.to(xbuf.device)
                dist.broadcast(uid, src=0, group=group)
                uid = uid.cpu()
            raw = (C.c_uint8 * 128)(*uid.tolist())
            self._check(self._lib.dompc_rccl_init(self._h, path, raw, rank, world))
        cb = _ALLREDUCE_FN() if native else _ALLREDUCE_FN(callback)
        arrays = {k: np.ascontiguousarray(t[k]) for k in ("x_mask", "g_mask", "edge_mask", "node_mask", "node_cut")}
        d = ShardDesc(rank=rank, world=world, cut_level=t["cut_level"], n_cut=t["n_cut"], xbuf=base, xbuf_doubles=n, allreduce=cb, ctx=None,
                      **{k: v.ctypes.data for k, v in arrays.items()})
        self._check(self._lib.dompc_set_sharding(self._h, C.byref(d)))
        self._shard = {"tables": t, "xbuf": xbuf, "callback": cb, "reduce": reduce_view, "stream": stream, "arrays": arrays,
                       "native_rccl": native}
        if not native or world > 1:
            reduce_view(xbuf[:1])                                      # communicator warm-up outside the solve
        return t

    def disable_sharding(self):
        self._check(self._lib.dompc_set_sharding(self._h, None))
        self._shard = None

    def _sum_over_ranks(self, a: np.ndarray) -> np.ndarray:
        import torch
        if self._shard["tables"]["world"] == 1:
            return a
        if self._host_emulation:
            v = torch.from_numpy(np.ascontiguousarray(a))
            self._shard["reduce"](v)
            return v.numpy()
        v = torch.from_numpy(np.ascontiguousarray(a)).to(self._shard["xbuf"].device)
        self._shard["reduce"](v)
        return v.cpu().numpy()

    def _stats_dict(self, s) -> dict:
        status = self._lib.dompc_status_string(int(s["status"])).decode()
        return {"success": bool(s["success"]), "return_status": status, "iter_count": int(s["iter_count"]),
                "t_wall_total": float(s["t_wall_total"]), "t_proc_total": float(s["t_wall_total"]),
                "n_reg": int(s["n_reg"]), "n_ls_fail": int(s["n_ls_fail"]), "n_sweeps": int(s["n_sweeps"]), "n_trials": int(s["n_trials"]), "n_soc": int(s["n_soc"]), "n_watchdog": int(s["n_watchdog"]),
                "mu": float(s["mu"]), "obj": float(s["obj"]), "inf_pr": float(s["inf_pr"]),
                "inf_du": float(s["inf_du"]), "obj_scaling": float(s["obj_scaling"]),
                "unified_return_status": "SOLVER_RET_SUCCESS" if s["success"] else "SOLVER_RET_UNKNOWN"}

    def stats(self) -> dict:
        return dict(self._stats)

    @property
    def code_object_info(self) -> dict:
        """Which gfx950 code objects this handle runs and how they were compiled: the machine-scheduler flags each one got (build.py falls
        back to the default strategy for a model on which this ROCm's clang crashes with the iterative scheduler - `<object>.sched` records
        it), whether the launch-shape-specific `_batch` sibling is in use, and the edges per wavefront of the derivative sweep."""
        def sched_of(path):
            try:
                with open(path + ".sched") as f:
                    return f.read().strip() or "default"
            except OSError:
                return "unknown"
        co = self.code_object_path
        info = {"object": co, "sched": sched_of(co) if co else "unknown", "edges_per_wavefront": self.edges_per_wavefront,
                "batch_object_state": self.batch_object_state}
        if co and self.batch_object_state == 1:
            info["batch_object"] = co[:-len(".hsaco")] + "_batch.hsaco"
            info["batch_sched"] = sched_of(info["batch_object"])
        return info

    def abort(self, stop: bool = True):
        """Stop request (dompc_abort): solves in flight leave their IPM loop with return_status
        'User_Requested_Stop'; `abort(False)` re-arms the handle.  May be called from another thread."""
        self._check(self._lib.dompc_abort(self._h, 1 if stop else 0))

    # ------------------------------------------------------------------ batch (host buffers)
    def solve_batch(self, X0, lbx, ubx, lbg, ubg, P):
        ps = self.structure
        X0 = np.ascontiguousarray(X0, dtype=np.float64).reshape(-1, ps.n_opt_x)
        P = np.ascontiguousarray(P, dtype=np.float64).reshape(-1, ps.n_opt_p)
        B = X0.shape[0]
        assert P.shape[0] == B
        lbx, ubx, lbg, ubg = _f64(lbx, ps.n_opt_x), _f64(ubx, ps.n_opt_x), _f64(lbg, ps.n_g), _f64(ubg, ps.n_g)
        X = np.empty((B, ps.n_opt_x))
        G = np.empty((B, ps.n_g))
        LX = np.empty((B, ps.n_opt_x))
        LG = np.empty((B, ps.n_g))
        F = np.empty(B)
        st = np.zeros(B, dtype=STATS_DTYPE)
        self._check(self._lib.dompc_solve_batch(self._h, B, _ptr(X0), _ptr(lbx), _ptr(ubx), _ptr(lbg), _ptr(ubg),
                                                _ptr(P), _ptr(X), _ptr(G), _ptr(LX), _ptr(LG), _ptr(F), _ptr(st)))
        return {"x": X, "f": F, "g": G, "lam_x": LX, "lam_g": LG, "stats": st}

    # ------------------------------------------------------------------ batch (device pointers)
    def solve_batch_device(self, B, x0, lbx, ubx, lbg, ubg, p, x, g, lam_x, lam_g, f, stats, stream=0):
        """All arguments are raw device addresses (ints, e.g. torch tensor .data_ptr()); asynchronous."""
        args = [C.c_void_p(int(a) if a else None) for a in (x0, lbx, ubx, lbg, ubg, p, x, g, lam_x, lam_g, f, stats)]
        self._check(self._lib.dompc_solve_batch_device(self._h, int(B), *args, C.c_void_p(int(stream) if stream else None)))

    def sweep_batch_device(self, B, x, lam, p, g, blocks, stream=0):
        args = [C.c_void_p(int(a)) for a in (x, lam, p, g, blocks)]
        self._check(self._lib.dompc_sweep_batch_device(self._h, int(B), *args, C.c_void_p(int(stream) if stream else None)))

    @property
    def sweep_block_doubles(self) -> int:
        return int(self._lib.dompc_sweep_block_doubles(self._h))

    @property
    def workspace_bytes(self) -> int:
        return int(self._lib.dompc_workspace_bytes(self._h))

    @property
    def num_slots(self) -> int:
        return int(self._lib.dompc_num_slots(self._h))

    def trace(self, n_rows: int) -> np.ndarray:
        out = np.zeros((max(int(n_rows), 1), 8))
        self._check(self._lib.dompc_debug_get_trace(self._h, _ptr(out), out.shape[0]))
        return out[:n_rows]

    # ------------------------------------------------------------------ parity hook
    def newton_step_at_solution(self, x, lam_g, zl, zu, lbx, ubx, lbg, ubg, p, mu):
        """Newton direction (dx, dlam) of the primal-dual system at a converged point of the barrier problem
        (`dompc_newton_step_at_solution`: slack variables of the nl_cons rows at their values of the point)."""
        ps = self.structure
        a = [_f64(v) for v in (x, lam_g, zl, zu, lbx, ubx, lbg, ubg, p)]
        dx = np.empty(ps.n_opt_x)
        dlam = np.empty(ps.n_g)
        self._check(self._lib.dompc_newton_step_at_solution(self._h, *[_ptr(v) for v in a], float(mu), _ptr(dx), _ptr(dlam)))
        return dx, dlam

    def newton_steps_at_solution(self, x, lam_g, zl, zu, lbx, ubx, lbg, ubg, P, mu):
        """`dompc_newton_steps_at_solution`: the Newton directions for B parameter vectors (rows of P) at one point, one launch"""
        ps = self.structure
        P = np.ascontiguousarray(np.asarray(P, dtype=np.float64).reshape(-1, ps.n_opt_p))
        B = P.shape[0]
        a = [_f64(v) for v in (x, lam_g, zl, zu, lbx, ubx, lbg, ubg)]
        dx = np.empty((B, ps.n_opt_x))
        dlam = np.empty((B, ps.n_g))
        self._check(self._lib.dompc_newton_steps_at_solution(self._h, B, *[_ptr(v) for v in a], _ptr(P), float(mu), _ptr(dx), _ptr(dlam)))
        return dx, dlam

    def debug_newton_step(self, x, lam_g, zl, zu, lbx, ubx, lbg, ubg, p, mu, delta_w=0.0):
        ps = self.structure
        a = [_f64(v) for v in (x, lam_g, zl, zu, lbx, ubx, lbg, ubg, p)]
        dx = np.empty(ps.n_opt_x)
        dlam = np.empty(ps.n_g)
        rd = np.empty(ps.n_opt_x)
        c = np.empty(ps.n_g)
        self._check(self._lib.dompc_debug_newton_step(self._h, *[_ptr(v) for v in a], float(mu), float(delta_w),
                                                      _ptr(dx), _ptr(dlam), _ptr(rd), _ptr(c)))
        return dx, dlam, rd, c



class RowMappedSolver:
    """HipIpmSolver of an NLP with node-local inequality rows appended to `nlp_cons` (nlp_route.ConstraintExtras), behind the reference's row
    order: the inner solver works on the internal layout (every edge has extra row slots), this wrapper scatters lbg / ubg into it (masked
    slots: -inf / +inf) and gathers g / lam_g back - structured rows first, then the appended rows in the order they were appended
    (optimizer.py:1086-1094).  Host-array entry points only; what needs the inner layout (device-resident batches, the differentiator, tree
    sharding) asks `row_mapped` and refuses."""

    row_mapped = True

    def __init__(self, inner: HipIpmSolver, row_map: np.ndarray):
        self.inner = inner
        self.row_map = np.asarray(row_map, np.int64)
        self.structure = getattr(inner, "structure", None)      # (internal layout)
        self.n_g_ref = self.row_map.size
        # the map is the identity on runs of rows (the initial-condition rows, then one run per edge, then the appended rows one by one): batches
        # are gathered run by run with slice copies - fancy indexing of a (B x n_g) array costs seconds at B = 16 384
        m = self.row_map
        cuts = np.flatnonzero(np.diff(m) != 1) + 1
        starts = np.concatenate([[0], cuts])
        ends = np.concatenate([cuts, [m.size]])
        self._runs = [(int(a), int(m[a]), int(b - a)) for a, b in zip(starts, ends)]      # (first reference row, first internal row, length)

    def __getattr__(self, name):
        if name in ("solve_batch_device", "sweep_batch_device", "enable_sharding", "newton_step", "newton_steps_at_solution", "debug_newton_step"):
            raise NotImplementedError("structured HIP backend: %s with node-local rows appended to nlp_cons (internal row layout); "
                                      "use make_step / make_step_batch" % name)
        return getattr(self.inner, name)

    def _bounds(self, lbg, ubg):
        n = self.structure.n_g
        lo, hi = np.full(n, -np.inf), np.full(n, np.inf)
        lo[self.row_map] = np.asarray(lbg, float).reshape(-1)
        hi[self.row_map] = np.asarray(ubg, float).reshape(-1)
        return lo, hi

    def __call__(self, x0, lbx, ubx, lbg, ubg, p, lam_x0=None, lam_g0=None) -> dict:
        lo, hi = self._bounds(lbg, ubg)
        r = self.inner(x0=x0, lbx=lbx, ubx=ubx, lbg=lo, ubg=hi, p=p)
        r = dict(r)
        r["g"], r["lam_g"] = r["g"][self.row_map], r["lam_g"][self.row_map]
        return r

    def solve_batch(self, X0, lbx, ubx, lbg, ubg, P):
        lo, hi = self._bounds(lbg, ubg)
        t0 = time.perf_counter()
        r = self.inner.solve_batch(X0, lbx, ubx, lo, hi, P)
        t1 = time.perf_counter()
        for key in ("g", "lam_g"):
            src = r[key]
            dst = np.empty((src.shape[0], self.n_g_ref))
            for d0, s0, n in self._runs:
                dst[:, d0:d0 + n] = src[:, s0:s0 + n]
            r[key] = dst
        if os.environ.get("DOMPC_TIMING"):
            print("[dompc timing] RowMappedSolver.solve_batch: inner %.3f s, rows back into the reference's order %.3f s" % (t1 - t0, time.perf_counter() - t1), file=sys.stderr)
        return r

    def stats(self):
        return self.inner.stats()

    def close(self):
        self.inner.close()
