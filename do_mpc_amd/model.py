"""Symbolic model container with the same user surface as do_mpc.model.Model.

Mirrors /root/reference/do_mpc/model/_model.py:
  __init__            :91-128     (variable groups incl. the zero-size 'default' entries)
  set_variable        :537-621
  set_expression      :623-668
  set_meas            :670-747
  set_rhs             :749-809
  set_alg             :811-841
  setup               :937-1058
but is built on do_mpc_amd.sym (our own scalar DAG) instead of CasADi structs, because
the expressions are lowered to HIP device functions at MPC.setup() rather than being
interpreted by CasADi's VM.
"""
from __future__ import annotations

from typing import Dict, List, Tuple, Union

import numpy as np

from . import sym
from .structs import Entry, Layout, NumStruct


class VarGroup:
    """Ordered {name: SX} with the struct-ish accessors user code relies on
    (model.x['m_P'], model.aux['cost'], model._u.keys(), .cat, .labels(), (0) -> numeric)."""

    def __init__(self, kind: str):
        self.kind = kind
        self.names: List[str] = []
        self.vars: Dict[str, sym.SX] = {}

    def add(self, name: str, var: sym.SX):
        if name in self.vars:
            raise Exception(f"The variable {name} for type {self.kind} already exists.")
        self.names.append(name)
        self.vars[name] = var

    def keys(self):
        return list(self.names)

    def __contains__(self, name):
        return name in self.vars

    def __getitem__(self, key) -> sym.SX:
        if isinstance(key, tuple):
            name, rest = key[0], key[1:]
            v = self.vars[name]
            return v[rest if len(rest) > 1 else rest[0]]
        return self.vars[key]

    @property
    def cat(self) -> sym.SX:
        parts = [self.vars[n].reshape((self.vars[n].numel(), 1)) for n in self.names]
        return sym.vertcat(*parts) if parts else sym.SX([], (0, 1))

    @property
    def size(self) -> int:
        return sum(self.vars[n].numel() for n in self.names)

    @property
    def shape(self):
        return (self.size, 1)

    def layout(self) -> Layout:
        return Layout([Entry(n, self.vars[n].shape) for n in self.names])

    def labels(self):
        return self.layout().labels()

    def offset(self, name: str) -> int:
        off = 0
        for n in self.names:
            if n == name:
                return off
            off += self.vars[n].numel()
        raise KeyError(name)

    def __call__(self, value=0.0) -> NumStruct:
        return NumStruct(self.layout(), value)


_VAR_TYPES = ("_x", "_u", "_z", "_p", "_tvp", "_w", "_v")
_LONG_VAR_TYPES = {"states": "_x", "inputs": "_u", "algebraic": "_z", "parameter": "_p", "timevarying_parameter": "_tvp"}


class Model:
    def __init__(self, model_type: str = None, symvar_type: str = "SX"):
        assert isinstance(model_type, str), "model_type must be string, you have: {}".format(type(model_type))
        assert model_type in ["discrete", "continuous"], \
            "model_type must be either discrete or continuous, you have: {}".format(model_type)
        assert symvar_type in ["SX", "MX"], "symvar_type must be either SX or MX, you have: {}".format(symvar_type)
        # 'MX' is accepted for source compatibility; both map onto the same scalar DAG here.
        self.symvar_type = symvar_type
        self.model_type = model_type
        self._x = VarGroup("_x")
        self._u = VarGroup("_u")
        self._z = VarGroup("_z")
        self._p = VarGroup("_p")
        self._tvp = VarGroup("_tvp")
        self._w = VarGroup("_w")
        self._v = VarGroup("_v")
        self._y = VarGroup("_y")
        self._aux = VarGroup("_aux")
        for grp in (self._u, self._z, self._p, self._tvp, self._w, self._v):
            grp.add("default", sym.SX([], (0, 0)))
        self._aux.add("default", sym.SX(0.0))            # size-1 'default' = 0  (_model.py:107,116)
        self._y_noise: Dict[str, bool] = {}
        self.rhs_list: List[dict] = []
        self.alg_list: List[dict] = []
        self.integer: List[str] = []
        self.flags = {"setup": False}

    # ---------------------------------------------------------------- queries
    def __getitem__(self, ind):
        """`model['x']`, `model['x', 'tvp']`: the variable groups themselves (name-indexable, `.cat` for the vector),
        like the structures the reference hands out (_model.py:165-200)."""
        if isinstance(ind, tuple):
            return [self._getvar(i) for i in ind]
        return self._getvar(ind)

    def _getvar(self, var_name: str) -> VarGroup:
        if var_name.startswith("_"):
            var_name = var_name[1:]
        table = {"x": self._x, "u": self._u, "z": self._z, "p": self._p, "tvp": self._tvp,
                 "y": self._y, "aux": self._aux, "w": self._w, "v": self._v}
        if var_name not in table:
            raise Exception(f"{var_name} is not a model variable type")
        return table[var_name]

    x = property(lambda self: self._x)
    u = property(lambda self: self._u)
    z = property(lambda self: self._z)
    p = property(lambda self: self._p)
    tvp = property(lambda self: self._tvp)
    y = property(lambda self: self._y)
    aux = property(lambda self: self._aux)
    w = property(lambda self: self._w)
    v = property(lambda self: self._v)

    n_x = property(lambda self: self._x.size)
    n_u = property(lambda self: self._u.size)
    n_z = property(lambda self: self._z.size)
    n_p = property(lambda self: self._p.size)
    n_tvp = property(lambda self: self._tvp.size)
    n_w = property(lambda self: self._w.size)
    n_v = property(lambda self: self._v.size)
    n_y = property(lambda self: self._y.size)
    n_aux = property(lambda self: self._aux.size)

    # ---------------------------------------------------------------- configuration
    def set_variable(self, var_type: str, var_name: str, shape: Union[int, Tuple] = (1, 1),
                     input_type_integer: bool = False) -> sym.SX:
        assert self.flags["setup"] is False, "Cannot call .set_variable after setup."
        assert isinstance(var_type, str), "var_type must be str, you have: {}".format(type(var_type))
        assert isinstance(var_name, str), "var_name must be str, you have: {}".format(type(var_name))
        assert isinstance(shape, (tuple, int)), "shape must be tuple or int, you have: {}".format(type(shape))
        var_type = _LONG_VAR_TYPES.get(var_type, var_type)          # long names (_model.py:596-601)
        if var_type not in _VAR_TYPES:
            raise Exception("Trying to set non-existing variable var_type: {} with var_name {}".format(var_type, var_name))
        if isinstance(shape, int):
            shape = (shape, 1)
        if var_type != "_u" and input_type_integer:
            raise Exception("Integer variables are only supported for inputs (_u).")
        grp = getattr(self, var_type)
        var = sym.SX.sym(var_name, shape[0], shape[1])
        grp.add(var_name, var)
        if input_type_integer:
            self.integer.append(var_name)
        return var

    def set_expression(self, expr_name: str, expr) -> sym.SX:
        assert self.flags["setup"] is False, "Cannot call .set_expression after setup."
        assert isinstance(expr_name, str), "expr_name must be str, you have: {}".format(type(expr_name))
        assert isinstance(expr, (sym.SX, sym.DM)), "expr must be a symbolic expression, you have: {}".format(type(expr))
        expr = sym.SX(expr)
        self._aux.add(expr_name, expr)
        return expr

    def set_meas(self, meas_name: str, expr, meas_noise: bool = True) -> sym.SX:
        assert self.flags["setup"] is False, "Cannot call .set_meas after setup."
        assert isinstance(meas_name, str), "meas_name must be str, you have: {}".format(type(meas_name))
        expr = sym.SX(expr)
        if meas_noise:
            v = self.set_variable("_v", meas_name + "_noise", expr.shape)
            expr = expr + v
        self._y.add(meas_name, expr)
        return expr

    def set_rhs(self, var_name: str, expr, process_noise: bool = False) -> None:
        assert self.flags["setup"] is False, "Cannot call .set_rhs after .setup."
        assert isinstance(var_name, str), "var_name must be str, you have: {}".format(type(var_name))
        assert var_name in self._x.names, \
            "var_name must refer to the previously defined states ({}). You have: {}".format(self._x.names, var_name)
        expr = sym.SX(expr)
        if process_noise:
            w = self.set_variable("_w", var_name + "_noise", expr.shape)
            expr = expr + w
        self.rhs_list.append({"var_name": var_name, "expr": expr})

    def set_alg(self, expr_name: str, expr) -> None:
        assert self.flags["setup"] is False, "Cannot call .set_alg after .setup."
        self.alg_list.append({"expr_name": expr_name, "expr": sym.SX(expr)})

    # ---------------------------------------------------------------- finalise
    def setup(self) -> None:
        if len(self._y.names) == 0:           # default: full state feedback (_model.py:954-957)
            for n in self._x.names:
                self._y.add(n, self._x.vars[n])
        rhs_names = [r["var_name"] for r in self.rhs_list]
        for n in self._x.names:
            if n not in rhs_names:
                raise Exception(f"Set rhs for all states. Missing: {n}")
        for r in self.rhs_list:
            if r["expr"].shape != self._x.vars[r["var_name"]].shape:
                raise Exception(f"rhs for {r['var_name']} has shape {r['expr'].shape}, "
                                f"expected {self._x.vars[r['var_name']].shape}")
        by_name = {r["var_name"]: r["expr"] for r in self.rhs_list}
        self._rhs = sym.vertcat(*[by_name[n].reshape((by_name[n].numel(), 1)) for n in self._x.names])
        self._alg = sym.vertcat(*[a["expr"].reshape((a["expr"].numel(), 1)) for a in self.alg_list]) \
            if self.alg_list else sym.SX([], (0, 1))
        if self._alg.numel() != self.n_z:
            raise Exception(f"{self.n_z} algebraic states but {self._alg.numel()} algebraic equations")
        _x, _u, _z, _tvp, _p, _w, _v = (self._getvar(k).cat for k in ("x", "u", "z", "tvp", "p", "w", "v"))
        self._rhs_fun = sym.Function("rhs_fun", [_x, _u, _z, _tvp, _p, _w], [self._rhs])
        self._alg_fun = sym.Function("alg_fun", [_x, _u, _z, _tvp, _p, _w], [self._alg])
        self._aux_expression_fun = sym.Function("aux_expression_fun", [_x, _u, _z, _tvp, _p], [self._aux.cat])
        self._meas_fun = sym.Function("meas_fun", [_x, _u, _z, _tvp, _p, _v], [self._y.cat])
        for fn, what in ((self._rhs_fun, "rhs"), (self._aux_expression_fun, "aux")):
            free = fn.free_symbols()
            if free:
                raise Exception(f"{what} depends on symbols that are not model variables: {free}")
        self.flags["setup"] = True
