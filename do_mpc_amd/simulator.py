"""Simulator - the plant side of the closed loop, batched on the GPU.

Mirror of the reference's `do_mpc.simulator.Simulator` surface (/root/reference/do_mpc/simulator.py:106-850):
`Simulator(model)`, `settings.t_step / abstol / reltol / integration_tool` (`set_param(**kw)` forwards to it),
`get_p_template / set_p_fun`, `get_tvp_template / set_tvp_fun`, `setup()`, the iterated variables `x0`, `u0`, `t0`,
`make_step(u0, v0=None, w0=None) -> y_next`.  Underneath, the CVODES integrator object of simulator.py:381-416 is
replaced by the batched Runge-Kutta integrator of csrc/dompc_plant.hip (explicit pair + implicit SDIRK) behind the C ABI `dompc_plant_*`
(include/dompc_ipm.h); `make_step_batch(X, U, ...)` advances B samples with one launch, and
`step_batch_device(...)` does the same on device pointers so that an x0 batch never leaves HBM between the
controller's `make_step_batch` calls.

There is no CPU fallback: without a HIP device `setup()` raises.  Stiff plants: `settings.integration_tool` ('cvodes' / 'idas':
explicit pair with an implicit repeat for stiff samples; 'sdirk4': implicit only; 'dopri5': explicit only).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Callable, Dict, Optional

import numpy as np

from . import build, lowering, sym
from .model import Model
from .structs import Entry, Layout, NumStruct


@dataclass
class SimulatorSettings:
    """settings of the reference's ContinousSimulatorSettings (simulator.py:42-104)"""
    t_step: float = None
    abstol: float = 1e-10
    reltol: float = 1e-10
    # 'cvodes' / 'idas' (the reference's implicit integrators): explicit pair, stiff samples repeat their interval with the
    # implicit SDIRK method; 'dopri5': explicit only; 'sdirk4': implicit only (csrc/dompc_plant.hip)
    integration_tool: str = "cvodes"
    integration_opts: Dict = field(default_factory=dict)
    gpu_index: int = 0
    max_steps: int = 0                    # integration steps per sample and control interval (0 = 200000)

    def check_for_mandatory_settings(self):
        if self.t_step is None:
            raise ValueError("t_step must be set")


class PlantDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("nx", "nu", "np", "ntvp", "nw", "nv", "ny", "discrete")] + \
               [("code_object_path", C.c_char_p), ("model_hash", C.c_char_p), ("device", C.c_int32), ("max_steps", C.c_int32),
                ("t_step", C.c_double), ("reltol", C.c_double), ("abstol", C.c_double)]


def _bind(lib_path: str) -> C.CDLL:
    lib = C.CDLL(lib_path)
    vp = C.c_void_p
    lib.dompc_plant_create.argtypes = [C.POINTER(PlantDesc), C.POINTER(vp)]
    lib.dompc_plant_create.restype = C.c_int
    lib.dompc_plant_destroy.argtypes = [vp]
    lib.dompc_plant_last_error.argtypes = [vp]
    lib.dompc_plant_last_error.restype = C.c_char_p
    lib.dompc_plant_step_batch.argtypes = [vp, C.c_int32] + [vp] * 6 + [C.c_int32] + [vp] * 3
    lib.dompc_plant_step_batch.restype = C.c_int
    lib.dompc_plant_step_batch_device.argtypes = [vp, C.c_int32] + [vp] * 6 + [C.c_int32] + [vp] * 3 + [vp]
    lib.dompc_plant_step_batch_device.restype = C.c_int
    lib.dompc_plant_set_method.argtypes = [vp, C.c_int32, C.c_int32]
    lib.dompc_plant_set_method.restype = C.c_int
    lib.dompc_plant_set_z0.argtypes = [vp, vp]
    lib.dompc_plant_set_z0.restype = C.c_int
    lib.dompc_plant_set_z_carry.argtypes = [vp, C.c_int32]
    lib.dompc_plant_set_z_carry.restype = C.c_int
    lib.dompc_plant_num_alg_states.argtypes = [vp]
    lib.dompc_plant_num_alg_states.restype = C.c_int32
    return lib


def _rows(a, n: int, B: int):
    """-> (contiguous f64 array or None, shared flag): one row of n values shared by the batch, or [B][n]"""
    if n == 0:
        return np.zeros(1), True
    if a is None:
        return np.zeros(n), True
    if hasattr(a, "master"):
        a = a.master
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    if a.size == n:
        return a.reshape(n), True
    if a.shape != (B, n):
        raise ValueError(f"expected {n} values or an array of shape ({B}, {n}), got {a.shape}")
    return a, False


def _flat_struct(v, n):
    return np.asarray(v.master if hasattr(v, "master") else v, dtype=float).reshape(-1)[:n] if n else np.zeros(0)


class Simulator:
    def __init__(self, model: Model):
        assert model.flags["setup"] is True, "Model for simulator was not setup. After the complete model creation call model.setup()."
        self.model = model
        self.settings = SimulatorSettings()
        if model.model_type == "discrete":
            self.settings.t_step = self.settings.t_step
        self._x0 = model._x(0.0)
        self._u0 = model._u(0.0)
        self._z0 = model._z(0.0)
        self._t0 = np.array([0.0])
        from .controller import MPCData                 # per-step records like the reference's simulator.data (simulator.py:833-841)
        self.data = MPCData(model)
        self.flags = {"set_tvp_fun": False, "set_p_fun": False, "setup": False, "first_step": True}
        self._h = None
        self._lib = None

    # ------------------------------------------------------------------ iterated variables (model/_iteratedvariables.py)
    def _set_iter(self, name, v):
        tgt = getattr(self, name)
        a = np.asarray(v.master if hasattr(v, "master") else v, dtype=float).reshape(-1)
        assert a.size == tgt.master.size, f"{name} has incorrect size {a.size}, expected {tgt.master.size}"
        tgt.master[:] = a

    x0 = property(lambda self: self._x0, lambda self, v: self._set_iter("_x0", v))
    z0 = property(lambda self: self._z0, lambda self, v: self._set_iter("_z0", v))
    u0 = property(lambda self: self._u0, lambda self, v: self._set_iter("_u0", v))
    t0 = property(lambda self: self._t0)

    # ------------------------------------------------------------------ configuration
    def set_param(self, **kwargs) -> None:
        for k, v in kwargs.items():
            if not hasattr(self.settings, k):
                print(f"Warning: Key {k} does not exist for Simulator.")
            else:
                setattr(self.settings, k, v)

    def get_tvp_template(self) -> NumStruct:
        return self.model._tvp(0.0)

    def set_tvp_fun(self, tvp_fun: Callable) -> None:
        assert self.get_tvp_template().labels() == tvp_fun(0).labels(), \
            "Incorrect output of tvp_fun. Use get_tvp_template to obtain the required structure."
        self.tvp_fun = tvp_fun
        self.flags["set_tvp_fun"] = True

    def get_p_template(self) -> NumStruct:
        return self.model._p(0.0)

    def set_p_fun(self, p_fun: Callable) -> None:
        assert self.get_p_template().labels() == p_fun(0).labels(), \
            "Incorrect output of p_fun. Use get_p_template to obtain the required structure."
        self.p_fun = p_fun
        self.flags["set_p_fun"] = True

    def _check_validity(self):
        # simulator.py:299-319: default (constant zero) parameter functions when the model has none
        if not self.flags["set_tvp_fun"]:
            if self.model.n_tvp:
                raise Exception("You have not supplied a function to obtain the time-varying parameters defined in model. "
                                "Use .set_tvp_fun() prior to setup.")
            tvp0 = self.get_tvp_template()
            self.tvp_fun = lambda t: tvp0
        if not self.flags["set_p_fun"]:
            if self.model.n_p:
                raise Exception("You have not supplied a function to obtain the parameters defined in model. "
                                "Use .set_p_fun() prior to setup.")
            p0 = self.get_p_template()
            self.p_fun = lambda t: p0

    def _lower(self) -> str:
        m = self.model
        return lowering.lower_plant(
            x_sym=m._x.cat.nodes(), u_sym=m._u.cat.nodes(), tvp_sym=m._tvp.cat.nodes(), p_sym=m._p.cat.nodes(),
            w_sym=m._w.cat.nodes(), v_sym=m._v.cat.nodes(), rhs=m._rhs.nodes(), meas=m._y.cat.nodes(),
            discrete=m.model_type == "discrete", name=type(m).__name__,
            z_sym=m._z.cat.nodes(), alg=(m._alg.nodes() if m.n_z else []))

    def setup(self, _lib_path: Optional[str] = None, _code_object: Optional[str] = None) -> None:
        self.settings.check_for_mandatory_settings()
        self._check_validity()
        m = self.model
        self.generated_header = self._lower()
        self.model_hash = self.generated_header.rsplit('PLANT_MODEL_HASH "', 1)[1].split('"')[0]
        if _lib_path is None:
            import os
            if not os.environ.get("DOMPC_NO_TORCH_FIRST"):
                try:                      # torch ships its own HIP runtime: it has to be the first one in the process
                    import torch          # noqa: F401
                    torch.cuda.is_available()
                except ImportError:
                    pass
            _lib_path = build.runtime_library()
            _code_object = build.plant_code_object(self.generated_header, self.model_hash)
        self._lib = _bind(_lib_path)
        d = PlantDesc(nx=m.n_x, nu=m.n_u, np=m.n_p, ntvp=m.n_tvp, nw=m.n_w, nv=m.n_v, ny=m.n_y,
                      discrete=1 if m.model_type == "discrete" else 0,
                      code_object_path=(_code_object or "").encode(), model_hash=self.model_hash.encode(),
                      device=self.settings.gpu_index, max_steps=self.settings.max_steps,
                      t_step=float(self.settings.t_step), reltol=float(self.settings.reltol), abstol=float(self.settings.abstol))
        h = C.c_void_p()
        if self._lib.dompc_plant_create(C.byref(d), C.byref(h)) != 0:
            raise RuntimeError("dompc_plant_create failed: " + (self._lib.dompc_plant_last_error(None) or b"?").decode())
        self._h = h
        tool = str(self.settings.integration_tool).lower()
        methods = {"cvodes": 0, "idas": 0, "auto": 0, "dopri5": 1, "rk45": 1, "sdirk4": 2, "implicit": 2}
        if tool not in methods:
            raise ValueError(f"integration_tool {self.settings.integration_tool!r}: expected one of {sorted(methods)}")
        if self._lib.dompc_plant_set_method(h, methods[tool], int(self.settings.integration_opts.get("explicit_limit", 0))) != 0:
            raise RuntimeError("dompc_plant: " + (self._lib.dompc_plant_last_error(h) or b"?").decode())
        self._seed_z()
        self.flags["setup"] = True

    def _seed_z(self):
        """Newton start of the algebraic states on the device = simulator.z0 (simulator.py:603-620: sim_z_num)"""
        if self.model.n_z and self._h:
            z = np.ascontiguousarray(self._z0.master, dtype=np.float64)
            self._lib.dompc_plant_set_z0(self._h, z.ctypes.data_as(C.c_void_p))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.dompc_plant_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ runtime
    def set_initial_guess(self) -> None:
        """Initial guess of the algebraic states for the DAE solver (simulator.py:603-620): `simulator.z0` becomes the Newton start of
        the algebraic equations for every sample (the kernel then continues from the values it found at the end of the previous
        step, like IDAS from sim_z_num); the records' z (`_host_z`) starts from the same values."""
        assert self.flags["setup"], "Simulator was not setup yet. Please call Simulator.setup()."
        self._seed_z()

    def _host_z(self, x, u, tvp, p) -> np.ndarray:
        """algebraic states at (x, u) for the data records (Newton's method on the model's own functions, from the last values)"""
        m = self.model
        if not m.n_z:
            return np.zeros(0)
        if getattr(self, "_algJ_fun", None) is None:
            from . import sym
            ins = [m._x.cat, m._u.cat, m._z.cat, m._tvp.cat, m._p.cat, m._w.cat]
            self._algJ_fun = sym.Function("alg_jz", ins, [sym.jacobian(m._alg, m._z.cat)])
        z, w = self._z0.master.copy(), np.zeros(m.n_w)
        for _ in range(30):
            a = np.asarray(m._alg_fun.eval(x, u, z, tvp, p, w)[0], float).ravel()
            J = np.asarray(self._algJ_fun.eval(x, u, z, tvp, p, w)[0], float).reshape(m.n_z, m.n_z, order="F")
            dz = np.linalg.solve(J, a)
            z = z - dz
            if np.max(np.abs(dz)) <= 1e-13 * max(1.0, np.max(np.abs(z))):
                break
        self._z0.master[:] = z
        return z

    def make_step_batch(self, X, U=None, P=None, TVP=None, W=None, V=None, carry_z: bool = False) -> dict:
        """Advance B samples by one control interval.  X: [B][nx]; U, P, TVP, W, V: [B][n] or one row shared by all
        samples (P / TVP default to p_fun(t0) / tvp_fun(t0)).  Returns {'x', 'y', 'status', 'n_steps'}; status bit 0: the integration
        did not reach t_step (failure), bit 1: the implicit method produced the result (no failure).
        DAE plants: the Newton iteration for the algebraic states of every sample starts from `simulator.z0` - unless `carry_z`, which
        says that row b of this call continues the trajectory of row b of the previous call (then from the values found there)."""
        assert self.flags["setup"], "Simulator is not setup. Call simulator.setup() first."
        m = self.model
        X = np.ascontiguousarray(np.asarray(X, dtype=np.float64)).reshape(-1, m.n_x)
        B = X.shape[0]
        t0 = float(self._t0[0])
        u, su = _rows(U, m.n_u, B)
        p, sp = _rows(P if P is not None else self.p_fun(t0), m.n_p, B)
        tvp, st = _rows(TVP if TVP is not None else self.tvp_fun(t0), m.n_tvp, B)
        w, sw = _rows(W, m.n_w, B)
        v, sv = _rows(V, m.n_v, B)
        mask = (1 if su else 0) | (2 if st else 0) | (4 if sp else 0) | (8 if sw else 0) | (16 if sv else 0)
        xn = np.empty((B, m.n_x))
        y = np.empty((B, max(m.n_y, 1)))
        status = np.zeros(B, dtype=np.int32)
        ptr = lambda a: a.ctypes.data_as(C.c_void_p)      # noqa: E731
        self._lib.dompc_plant_set_z_carry(self._h, 1 if carry_z else 0)
        rc = self._lib.dompc_plant_step_batch(self._h, B, ptr(X), ptr(u), ptr(tvp), ptr(p), ptr(w), ptr(v), mask,
                                              ptr(xn), ptr(y), ptr(status))
        if rc != 0:
            raise RuntimeError("dompc_plant: " + (self._lib.dompc_plant_last_error(self._h) or b"?").decode())
        return {"x": xn, "y": y[:, :m.n_y], "status": status & 0xFF, "n_steps": status >> 8}

    def step_batch_device(self, B, x, u, tvp, p, x_next, y=0, status=0, w=0, v=0, shared_mask=0, stream=0):
        """All arguments are raw device addresses (ints, e.g. torch tensor .data_ptr()); asynchronous on `stream`."""
        args = [C.c_void_p(int(a) if a else None) for a in (x, u, tvp, p, w, v)]
        outs = [C.c_void_p(int(a) if a else None) for a in (x_next, y, status)]
        rc = self._lib.dompc_plant_step_batch_device(self._h, int(B), *args, int(shared_mask), *outs,
                                                     C.c_void_p(int(stream) if stream else None))
        if rc != 0:
            raise RuntimeError("dompc_plant: " + (self._lib.dompc_plant_last_error(self._h) or b"?").decode())

    def make_step(self, u0=None, v0=None, w0=None) -> np.ndarray:
        """One closed-loop sample (simulator.py:757-850): integrates x0 over t_step with u0 and the current p / tvp,
        stores the new state in x0 and returns the measurement y_next as a column vector."""
        assert self.flags["setup"], "Simulator is not setup. Call simulator.setup() first."
        m = self.model
        if u0 is None:
            assert m.n_u == 0, "No input u0 provided. Please provide an input u0."
            u0 = np.zeros((0, 1))
        u0 = np.asarray(u0.master if hasattr(u0, "master") else u0, dtype=float)
        assert u0.size == m.n_u, "u0 has incorrect shape. You have: {}, expected: {}".format(u0.shape, (m.n_u, 1))
        t0 = float(self._t0[0])
        p0 = _flat_struct(self.p_fun(t0), m.n_p)
        tvp0 = _flat_struct(self.tvp_fun(t0), m.n_tvp)
        r = self.make_step_batch(self._x0.master[None, :], U=u0.reshape(-1), P=p0, TVP=tvp0,
                                 W=None if w0 is None else np.asarray(w0, float).reshape(-1),
                                 V=None if v0 is None else np.asarray(v0, float).reshape(-1), carry_z=True)       # (one trajectory)
        if int(r["status"][0]) & 1:          # bit 0: failure; bit 1 only says that the implicit SDIRK method produced the result
            raise RuntimeError("plant integration did not reach t_step (step limit or NaN right-hand side)")
        # records of the step: state BEFORE the step, the inputs and parameters it used, the new measurement (simulator.py:833-841)
        z0 = self._host_z(self._x0.master, u0.reshape(-1), tvp0, p0)      # (records only: the kernel solves for z itself)
        aux0 = m._aux_expression_fun.eval(self._x0.master, u0.reshape(-1), z0, tvp0, p0)[0]
        self.data.update(_x=self._x0.master.copy(), _u=u0.reshape(-1), _z=z0, _tvp=tvp0, _p=p0, _y=r["y"][0],
                         _aux=np.asarray(aux0, float).reshape(-1), _time=self._t0.copy())
        self._x0.master[:] = r["x"][0]
        self._u0.master[:] = u0.reshape(-1)
        self._t0 = self._t0 + self.settings.t_step
        self.flags["first_step"] = False
        return r["y"][0].reshape(-1, 1)
