"""Minimal `casadi` / `casadi.tools` / `do_mpc` stand-ins so that the reference's shipped
`template_model.py` / `template_mpc.py` files run *un-edited* on this backend.

The templates start with `from casadi import *`, `from casadi.tools import *`, `import do_mpc`
(/root/reference/examples/industrial_poly/template_model.py:23-30).  CasADi is not installable in
the build/bench images, and on the hot path we do not want its VM anyway (the model is lowered to
gfx950 code), so `install()` registers small modules in `sys.modules` that expose exactly the names
those files touch, backed by do_mpc_amd.sym / do_mpc_amd.model / do_mpc_amd.controller.
With real CasADi installed do not call install(); use the ctypes stub of INTEGRATION.md instead.
"""
import sys
import types

from . import controller, differentiator, estimator, model, sampling, simulator, structs, sym

_CASADI_NAMES = [
    "SX", "DM", "vertcat", "horzcat", "vertsplit", "mtimes", "sum1", "sum2", "sumsqr", "dot", "exp", "log", "sqrt", "sin",
    "cos", "tan", "tanh", "sinh", "cosh", "asin", "acos", "atan", "atan2", "sign", "fabs", "fmin", "fmax", "jacobian",
    "gradient", "hessian", "substitute", "Function",
]


class StateFeedback:
    """`do_mpc.estimator.StateFeedback` (/root/reference/do_mpc/estimator/_base.py:55-72): passes the measurement
    through as the state estimate - what every closed loop of the shipped examples uses between simulator and controller."""

    def __init__(self, model):
        self.model = model
        self._x0 = model._x(0.0)
        self.data = controller.MPCData(model)

    x0 = property(lambda self: self._x0, lambda self, v: self._x0.master.__setitem__(slice(None), _flat(v)))

    def make_step(self, y0):
        self.data.update(_x=_flat(y0))
        return y0

    def reset_history(self):
        self.data.init_storage()


def _flat(v):
    import numpy as np
    return np.asarray(v.master if hasattr(v, "master") else v, dtype=float).reshape(-1)


def install(force: bool = False):
    """Register the stand-in modules.  Returns the list of module names that were installed."""
    installed = []
    if force or "casadi" not in sys.modules:
        cas = types.ModuleType("casadi")
        cas.__doc__ = "do_mpc_amd stand-in for the subset of CasADi used by do-mpc model/controller templates"
        for n in _CASADI_NAMES:
            setattr(cas, n, getattr(sym, n))
        cas.MX = sym.SX                      # both symbol flavours map onto the same scalar DAG
        cas.inf = float("inf")
        cas.pi = 3.141592653589793
        import os as _os
        cas.os, cas.sys = _os, sys           # (leak out of the real package's star import; triple_tank_ekf/template_model.py:25-27 relies on it)
        cas.__all__ = _CASADI_NAMES + ["MX", "inf", "pi", "os", "sys"]
        tools = types.ModuleType("casadi.tools")
        tools.entry = structs.entry
        tools.indexf = differentiator.indexf
        tools.__all__ = ["entry", "indexf"]
        cas.tools = tools
        sys.modules["casadi"] = cas
        sys.modules["casadi.tools"] = tools
        installed += ["casadi", "casadi.tools"]
    if force or "do_mpc" not in sys.modules:
        dm = types.ModuleType("do_mpc")
        dm.__doc__ = "do_mpc_amd stand-in exposing do_mpc.model.Model and do_mpc.controller.MPC"
        m_model = types.ModuleType("do_mpc.model")
        m_model.Model = model.Model
        m_ctrl = types.ModuleType("do_mpc.controller")
        m_ctrl.MPC = controller.MPC
        m_ctrl.MPCSettings = controller.MPCSettings
        m_sim = types.ModuleType("do_mpc.simulator")
        m_sim.Simulator = simulator.Simulator
        m_est = types.ModuleType("do_mpc.estimator")
        m_est.StateFeedback = StateFeedback
        m_est.MHE, m_est.MHESettings = estimator.MHE, estimator.MHESettings
        m_diff = types.ModuleType("do_mpc.differentiator")
        m_diff.DoMPCDifferentiator = differentiator.DoMPCDifferentiator
        m_samp = types.ModuleType("do_mpc.sampling")
        m_samp.SamplingPlanner, m_samp.Sampler, m_samp.DataHandler = (sampling.SamplingPlanner, sampling.Sampler,
                                                                      sampling.DataHandler)
        dm.model, dm.controller, dm.simulator, dm.estimator, dm.differentiator = m_model, m_ctrl, m_sim, m_est, m_diff
        dm.sampling = m_samp
        from . import data as _data
        m_data = types.ModuleType("do_mpc.data")
        m_data.save_results, m_data.load_results, m_data.MPCData = _data.save_results, _data.load_results, _data.MPCData
        dm.data = m_data
        m_ampc = types.ModuleType("do_mpc.approximateMPC")      # (the data-generation half; the torch model is out of scope)
        m_ampc.AMPCSampler = sampling.AMPCSampler
        dm.approximateMPC = m_ampc
        dm.__version__ = "5.1.1+dompc_amd"
        sys.modules["do_mpc"] = dm
        sys.modules["do_mpc.model"] = m_model
        sys.modules["do_mpc.controller"] = m_ctrl
        sys.modules["do_mpc.simulator"] = m_sim
        sys.modules["do_mpc.estimator"] = m_est
        sys.modules["do_mpc.differentiator"] = m_diff
        sys.modules["do_mpc.sampling"] = m_samp
        sys.modules["do_mpc.approximateMPC"] = m_ampc
        sys.modules["do_mpc.data"] = m_data
        installed += ["do_mpc", "do_mpc.model", "do_mpc.controller", "do_mpc.simulator", "do_mpc.estimator",
                      "do_mpc.differentiator", "do_mpc.sampling", "do_mpc.approximateMPC", "do_mpc.data"]
    return installed


def uninstall(names):
    for n in names:
        sys.modules.pop(n, None)
