"""Batched sampling drivers on top of `MPC.make_step_batch`.

What they replace in the reference: the per-sample loops that feed its data-driven tools -
`do_mpc.sampling.Sampler.sample_data` (/root/reference/do_mpc/sampling/_sampler.py:198-228: one call of the user's
sample function per entry of the sampling plan, fanned out over processes by the user) and, built on it, the open-loop
sampler of the approximate-MPC module (/root/reference/do_mpc/approximateMPC/_ampc_sampler.py:234-345: draw (x0, u_prev)
uniformly in a box, `mpc.reset_history(); mpc.x0 = x0; mpc.u0 = u_prev; mpc.set_initial_guess(); u0 = mpc.make_step(x0)` per
sample, collect u0 and the solver statistics into a table).  Here the whole plan is ONE device call: every sample is an
independent cold solve with the documented initial guess, i.e. exactly what `make_step_batch` does.
"""
import time
from typing import Dict, Optional

import numpy as np


def sampling_plan_box(lbx, ubx, lbu, ubu, n_samples: int, seed: Optional[int] = None) -> Dict[str, np.ndarray]:
    """(x0, u_prev) drawn uniformly in the box [lbx, ubx] x [lbu, ubu] (_ampc_sampler.py:234-273, `gen_x0` / `gen_u_prev`),
    as arrays instead of a list of per-sample dicts; `id` numbers the samples like the reference's planner."""
    rng = np.random.default_rng(seed)
    lbx, ubx, lbu, ubu = (np.asarray(v, float).ravel() for v in (lbx, ubx, lbu, ubu))
    return {"id": np.arange(n_samples),
            "x0": rng.uniform(lbx, ubx, size=(n_samples, lbx.size)),
            "u_prev": rng.uniform(lbu, ubu, size=(n_samples, lbu.size))}


def open_loop_samples(mpc, plan: Dict[str, np.ndarray], chunk: Optional[int] = None) -> Dict[str, np.ndarray]:
    """One cold `make_step` per entry of the plan, all entries of a chunk in one launch.  Returns the columns of the
    reference's result table (_ampc_sampler.py:322-345): x0, u_prev, u0, status (= solver success), iter_count,
    t_wall (kernel time of the launch divided by its samples), t_make_step (host wall time likewise)."""
    X0 = np.asarray(plan["x0"], float)
    UP = np.asarray(plan["u_prev"], float)
    n = X0.shape[0]
    chunk = n if not chunk else int(chunk)
    nu = mpc.model.n_u
    out = {"id": np.asarray(plan.get("id", np.arange(n))), "x0": X0, "u_prev": UP, "u0": np.zeros((n, nu)),
           "status": np.zeros(n, bool), "iter_count": np.zeros(n, int), "t_wall": np.zeros(n), "t_make_step": np.zeros(n)}
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        t0 = time.perf_counter()
        r = mpc.make_step_batch(X0[lo:hi], U_prev=UP[lo:hi])
        dt = time.perf_counter() - t0
        st = r["stats"]
        out["u0"][lo:hi] = r["u0"]
        out["status"][lo:hi] = st["success"] != 0
        out["iter_count"][lo:hi] = st["iter_count"]
        out["t_wall"][lo:hi] = float(np.max(st["t_wall_total"])) / (hi - lo)
        out["t_make_step"][lo:hi] = dt / (hi - lo)
    return out


def to_dataframe(samples: Dict[str, np.ndarray]):
    """The reference's `data_<name>_all.pkl` layout (one row per sample, array-valued cells for x0 / u_prev / u0)."""
    import pandas as pd
    n = len(samples["status"])
    return pd.DataFrame({k: ([v[i] for i in range(n)] if np.ndim(v) > 1 else v) for k, v in samples.items()})


def closed_loop_samples(mpc, simulator, plan: Dict[str, np.ndarray], trajectory_length: int, device: int = 0) -> Dict[str, np.ndarray]:
    """Closed-loop sampling of the approximate-MPC module (_ampc_sampler.py:384-470: per sample `trajectory_length` steps of
    mpc.make_step -> simulator.make_step -> state feedback from (x0, u_prev), stopped at the first failed solve) for the whole
    plan at once: controller and plant advance all samples together on the GPU (`BatchClosedLoop`).  Returns per sample the
    state trajectory x[T+1], the applied inputs u[T], the previous inputs u_prev[T] (the regression features of the
    reference: `u_prev_total`), the per-step solver success and `n_valid` = number of steps before the first failure."""
    from .closed_loop import BatchClosedLoop
    X0 = np.asarray(plan["x0"], float)
    UP = np.asarray(plan["u_prev"], float)
    n, T = X0.shape[0], int(trajectory_length)
    loop = BatchClosedLoop(mpc, simulator, X0, device=device, U_prev0=UP)
    x = np.zeros((n, T + 1, X0.shape[1]))
    u = np.zeros((n, T, UP.shape[1]))
    up = np.zeros((n, T, UP.shape[1]))
    ok = np.zeros((n, T), bool)
    iters = np.zeros((n, T), int)
    x[:, 0], cur_up = X0, UP.copy()
    import time as _time
    t_start = _time.perf_counter()
    for k in range(T):
        r = loop.step()
        up[:, k] = cur_up
        u[:, k], x[:, k + 1] = r["u0"], r["x"]
        ok[:, k] = (r["stats"]["success"] != 0) & (r["plant_status"] == 0)
        iters[:, k] = r["stats"]["iter_count"]
        cur_up = r["u0"]
    t_loop = _time.perf_counter() - t_start
    n_valid = np.where(ok.all(axis=1), T, np.argmin(ok, axis=1))
    return {"id": np.asarray(plan.get("id", np.arange(n))), "x": x, "u": u, "u_prev": up, "success": ok, "n_valid": n_valid,
            "iter_count": iters, "t_loop": t_loop}


# ----------------------------------------------------------------------------------------------------------------------
# The reference's sampling tool chain (`do_mpc.sampling`): plan -> samples on disk -> post-processed table.
# Same class and method names, file names and table layout as /root/reference/do_mpc/sampling/{_samplingplanner,_sampler,
# _datahandler}.py so that a plan written by one side is read by the other; pinned by the reference's own golden
# (testing/results/res_sampling_test_test_fun.pkl -> tests/golden/sampling_test_fun.json).  The one addition is
# `Sampler.set_batch_function`: a function that is handed every pending row of the plan AT ONCE (columns as arrays) and
# returns one result per row - the shape `open_loop_samples` / `make_step_batch` want: one launch for the whole plan
# instead of `n_samples` solver calls.
import copy as _copy
import inspect as _inspect
import itertools as _itertools
import logging as _logging
import os as _os
import pickle as _pickle


def _is_callable_or_none(f, allow_none=True):
    import types as _t
    return isinstance(f, (_t.FunctionType, _t.BuiltinFunctionType)) or (allow_none and f is None)


class _Settable:
    """`set_param(**kwargs)` over a fixed list of keys; unknown keys only warn (reference behaviour)."""
    _keys = ()

    def set_param(self, **kwargs) -> None:
        for k, v in kwargs.items():
            if k in self._keys:
                setattr(self, k, v)
            else:
                print("Warning: Key {} does not exist for {}.".format(k, type(self).__name__))

    @property
    def data_dir(self):
        return self._data_dir

    @data_dir.setter
    def data_dir(self, val):
        self._data_dir = val
        if self._make_dir:
            _os.makedirs(val, exist_ok=True)

    _make_dir = True

    def _sample_file(self, sample_id):
        ext = {"pickle": ".pkl", "mat": ".mat"}[self.save_format]
        return "{}{}_{}{}".format(self.data_dir, self.sample_name, sample_id, ext)


def _dump_pickle(path, obj):
    with open(path if path.endswith(".pkl") else path + ".pkl", "wb") as f:
        _pickle.dump(obj, f)


class SamplingPlanner(_Settable):
    """_samplingplanner.py:13-280.  A plan is a list of dicts {var: value, ..., 'id': zero-padded running number}."""
    _keys = ("overwrite", "id_precision")

    def __init__(self, **kwargs):
        self.sampling_vars, self.sampling_var_names, self.sampling_plan = [], [], []
        self.data_fields = list(self._keys)
        self.data_dir, self.overwrite, self.id_precision = "./", False, 3
        self.set_param(**kwargs)

    def set_sampling_var(self, name: str, fun_var_pdf=None) -> None:
        assert isinstance(name, str), "name must be str, you have {}".format(type(name))
        assert _is_callable_or_none(fun_var_pdf), "fun_var_pdf must be a function or None, you have {}".format(type(fun_var_pdf))
        self.sampling_vars.append({"name": name, "fun_var_pdf": fun_var_pdf})
        self.sampling_var_names.append(name)

    def add_sampling_case(self, **kwargs) -> list:
        unknown = [k for k in kwargs if k not in self.sampling_var_names]
        if unknown:
            raise Exception("{} is not a valid sampling variable. Introduce sampling variables with set_sampling_var."
                            .format(unknown[0]))
        case = dict(kwargs)                                   # given values first, drawn values after (order of the golden)
        for var in self.sampling_vars:
            if var["name"] not in kwargs:
                assert var["fun_var_pdf"] is not None, ("Cannot augment sampling_case for missing variable {}. Variable "
                                                        "generating function is missing.".format(var["name"]))
                case[var["name"]] = var["fun_var_pdf"]()
        case["id"] = str(len(self.sampling_plan)).zfill(self.id_precision)
        self.sampling_plan.append(case)
        return self.sampling_plan

    def gen_sampling_plan(self, n_samples: int) -> list:
        assert isinstance(n_samples, int), "n_samples must be int, you have {}".format(type(n_samples))
        assert n_samples > 0, "n_samples must be larger than 0."
        for _ in range(n_samples):
            self.add_sampling_case()
        return self.sampling_plan

    def product(self, **kwargs) -> list:
        if not all(isinstance(v, list) for v in kwargs.values()):
            raise ValueError("keyword values must be lists")
        if not all(k in self.sampling_var_names for k in kwargs):
            raise ValueError("keyword names must be existing sampling variables")
        for combo in _itertools.product(*kwargs.values()):
            self.add_sampling_case(**dict(zip(kwargs.keys(), combo)))
        return self.sampling_plan

    def export(self, sampling_plan_name: str) -> None:
        assert isinstance(sampling_plan_name, str), "sampling_plan_name must be of type str. You have {}.".format(
            type(sampling_plan_name))
        stem = self.data_dir + _os.path.splitext(sampling_plan_name)[0]
        suffixes = [""] if self.overwrite else _itertools.chain([""], map(str, range(1, 10000)))
        for s in suffixes:                                    # first free name unless overwriting
            if self.overwrite or not _os.path.isfile(stem + s + ".pkl"):
                _dump_pickle(stem + s + ".pkl", self.sampling_plan)
                return


class Sampler(_Settable):
    """_sampler.py:14-232: evaluates the sample function for every row of a plan and stores one file per row
    (`<sample_name>_<id>.pkl|.mat` under `data_dir`); rows whose file exists are skipped unless `overwrite`."""
    _keys = ("overwrite", "sample_name", "save_format", "print_progress")

    def __init__(self, sampling_plan: list, **kwargs):
        assert isinstance(sampling_plan, list), "sampling_plan must be a list"
        assert all(isinstance(r, dict) for r in sampling_plan), "All elements of sampling plan must be a dictionary."
        self.sampling_plan = sampling_plan
        self.sampling_vars = list(sampling_plan[0].keys())
        self.n_samples = len(sampling_plan)
        self.completion_list = []
        self.flags = {"set_sample_function": False}
        self.data_fields = list(self._keys)
        self.data_dir, self.sample_name, self.save_format = "./", "sample", "pickle"
        self.overwrite, self.print_progress, self.n_processes = False, True, 1
        self.sample_function = self.batch_function = None
        self.set_param(**kwargs)

    def _check_args(self, fun, what):
        assert _is_callable_or_none(fun, allow_none=False), what + " must be a function"
        extra = set(_inspect.getfullargspec(fun).args) - set(self.sampling_vars)
        assert not extra, ("{} must only contain keyword arguments that appear as sample vars in the sampling_plan. "
                           "You have the unknown arguments: {}".format(what, extra))

    def set_sample_function(self, sample_function) -> None:
        self._check_args(sample_function, "sample_function")
        self.sample_function = sample_function
        self.flags["set_sample_function"] = True

    def set_batch_function(self, batch_function) -> None:
        """`batch_function(**columns)`: every argument is the array of that sampling variable over the pending rows; returns
        a sequence with one result per row.  Used by `sample_data` instead of the row-by-row loop."""
        self._check_args(batch_function, "batch_function")
        self.batch_function = batch_function
        self.flags["set_sample_function"] = True

    def _pending(self, idx):
        return self.overwrite or not _os.path.isfile(self._sample_file(self.sampling_plan[idx]["id"]))

    def _store(self, idx, result):
        sid = self.sampling_plan[idx]["id"]
        name = self._sample_file(sid)
        if self.save_format == "pickle":
            _dump_pickle(name, result)
        else:
            import scipy.io as sio
            sio.savemat(name, {"res": result})
        self.completion_list.append(sid)

    def _progress(self):
        if self.print_progress:
            done, n = len(self.completion_list), max(1, self.n_samples)
            fill = int(50 * done // n)
            print("\rProgress: |{}{}| {:.1f}% Complete".format("█" * fill, "-" * (50 - fill), 100.0 * done / n),
                  end="\n" if done == n else "\r")

    def sample_idx(self, idx: int) -> None:
        assert self.flags["set_sample_function"], ("Cannot sample before setting the sample function with "
                                                   "Sampler.set_sample_function")
        assert 0 <= idx <= len(self.sampling_plan), "Invalid value for idx. Must be between 0 and {}. You have {}".format(
            len(self.sampling_plan), idx)
        if self._pending(idx):
            row = {k: v for k, v in self.sampling_plan[idx].items() if k != "id"}
            if self.sample_function is not None:
                result = self.sample_function(**row)
            else:
                result = self.batch_function(**{k: np.asarray([v]) for k, v in self._wanted(row).items()})[0]
            self._store(idx, result)
        self._progress()

    def _wanted(self, row):
        names = _inspect.getfullargspec(self.batch_function).args
        return {k: row[k] for k in names}

    def sample_data(self) -> None:
        if self.batch_function is None:
            for i in range(len(self.sampling_plan)):
                self.sample_idx(i)
            return
        todo = [i for i in range(len(self.sampling_plan)) if self._pending(i)]
        if todo:
            names = _inspect.getfullargspec(self.batch_function).args
            cols = {k: np.asarray([self.sampling_plan[i][k] for i in todo]) for k in names}
            results = self.batch_function(**cols)
            assert len(results) == len(todo), "batch_function must return one result per row of the plan"
            for i, r in zip(todo, results):
                self._store(i, r)
        self._progress()


class DataHandler(_Settable):
    """_datahandler.py:17-330: lazy loading of the stored samples + named post-processing functions; `dh[...]` and
    `dh.filter(input_filter, output_filter)` return lists of {plan row..., name: post-processed value...}."""
    _keys = ("data_dir", "sample_name", "save_format")
    _make_dir = False

    def __init__(self, sampling_plan, **kwargs):
        self.flags = {"set_post_processing": False}
        self.data_fields = list(self._keys)
        self.data_dir, self.sample_name, self.save_format = "./", "sample", "pickle"
        self.sampling_plan = sampling_plan
        self.sampling_vars = list(sampling_plan[0].keys())
        self.post_processing = {}
        self._cache = {}
        self.set_param(**kwargs)

    @property
    def pre_loaded_data(self):
        return {"id": list(self._cache.keys()), "data": list(self._cache.values())}

    def set_post_processing(self, name: str, post_processing_function) -> None:
        assert isinstance(name, str), "name must be str, you have {}".format(type(name))
        assert _is_callable_or_none(post_processing_function, allow_none=False), (
            "post_processing_function must be a function, you have {}".format(type(post_processing_function)))
        n_args = len(_inspect.signature(post_processing_function).parameters)
        self.post_processing[name] = {"function": post_processing_function, "n_args": n_args}
        self.flags["set_post_processing"] = True

    def _load(self, sample_id):
        name = self._sample_file(sample_id)
        try:
            if self.save_format == "pickle":
                with open(name, "rb") as f:
                    return _pickle.load(f)
            import scipy.io as sio
            return sio.loadmat(name)
        except FileNotFoundError:
            _logging.warning("Could not find or load file: {}. Check data_dir parameter and make sure sample has already "
                             "been generated.".format(name))
            return None

    def _result(self, row):
        if row["id"] not in self._cache:
            self._cache[row["id"]] = self._load(row["id"])
        return self._cache[row["id"]]

    def _table_row(self, row):
        result = self._result(row)
        out = _copy.copy(row)
        if not self.flags["set_post_processing"]:
            out["res"] = result
            return out
        for name, pp in self.post_processing.items():
            if result is None:
                out[name] = None
            elif pp["n_args"] == 1:
                out[name] = pp["function"](result)
            elif pp["n_args"] == 2:
                out[name] = pp["function"](row, result)
        return out

    def __getitem__(self, ind):
        if isinstance(ind, int):
            rows = [self.sampling_plan[ind]]
        elif isinstance(ind, slice):
            rows = self.sampling_plan[ind]
        elif isinstance(ind, (tuple, list)):
            rows = [self.sampling_plan[i] for i in ind]
        else:
            raise Exception("ind must be of type int, tuple, slice or list. You have {}".format(type(ind)))
        return [self._table_row(r) for r in rows]

    def filter(self, input_filter=None, output_filter=None) -> list:
        assert _is_callable_or_none(input_filter), "input_filter must be a function, you have {}".format(type(input_filter))
        assert _is_callable_or_none(output_filter), "output_filter must be a function, you have {}".format(type(output_filter))

        def passes(fun, values):
            if fun is None:
                return True
            names = fun.__code__.co_varnames[:fun.__code__.co_argcount]
            return fun(**{n: values[n] for n in names}) == True   # noqa: E712  (the filters may return numpy bools)

        out = []
        for row in self.sampling_plan:
            if passes(input_filter, row):
                t = self._table_row(row)
                if passes(output_filter, t):
                    out.append(t)
        return out


# ----------------------------------------------------------------------------------------------------------------------
# The data-generation half of the reference's approximate-MPC module (`do_mpc.approximateMPC.AMPCSampler`,
# /root/reference/do_mpc/approximateMPC/_ampc_sampler.py:41-526; its neural-network half is out of scope): same settings,
# same files (`<data_dir>/<name>/sampling_plan_<name>.pkl`, `samples_<name>/sample_<id>.pkl`, `data_<name>_all.pkl`,
# `data_<name>_opt.pkl`), but the plan is solved as batched launches instead of one `make_step` per row.
from dataclasses import dataclass as _dataclass, field as _field


@_dataclass
class SamplerSettings:
    """_ampcsettings.py:60-120."""
    n_samples: int = None
    dataset_name: str = None
    trajectory_length: int = None
    closed_loop_flag: bool = False
    data_dir: str = _os.path.join(".", "sampling")
    overwrite_sampler: bool = True
    lbx: list = None
    ubx: list = None
    lbu: list = None
    ubu: list = None
    lbp: list = None
    ubp: list = None
    chunk: int = None                      # (addition) rows per launch; default: the controller's `max_batch`

    def check_for_mandatory_settings(self):
        if self.n_samples is None:
            raise ValueError("n_samples must be set")
        if self.dataset_name is None:
            raise ValueError("dataset_name must be set")
        if self.closed_loop_flag and self.trajectory_length is None:
            raise ValueError("trajectory_length must be set for closed-loop sampling")


class AMPCSampler:
    def __init__(self, mpc, simulator=None):
        """`simulator`: the plant of the closed-loop sampling (a set-up `do_mpc_amd.simulator.Simulator`); by default one is
        built from the controller's model with its `t_step`, `p_fun` and `tvp_fun` like the reference does (n_robust = 0)."""
        self.mpc = mpc
        self._settings = SamplerSettings()
        st = self._settings
        st.lbx, st.ubx = mpc._x_lb.master.reshape(-1, 1).copy(), mpc._x_ub.master.reshape(-1, 1).copy()
        st.lbu, st.ubu = mpc._u_lb.master.reshape(-1, 1).copy(), mpc._u_ub.master.reshape(-1, 1).copy()
        self.simulator = simulator
        self.flags = {"setup": False}

    settings = property(lambda self: self._settings)

    def setup(self):
        assert self.flags["setup"] is False, "Setup can only be once."
        st = self._settings
        st.check_for_mandatory_settings()
        for nm, what in (("lbx", "lower bounds for state"), ("ubx", "upper bounds for state"),
                         ("lbu", "lower bounds for input"), ("ubu", "upper bounds for input")):
            assert not np.any(np.isinf(getattr(st, nm))), "There are missing {} variables that forms your sampling box.".format(what)
        if st.closed_loop_flag and self.simulator is None:
            if self.mpc.settings.n_robust != 0:
                raise NotImplementedError("AMPCSampler: closed-loop sampling with n_robust > 0 draws plant parameters per step "
                                          "(_ampc_sampler.py:431-440); pass a configured simulator instead")
            from .simulator import Simulator
            sim = Simulator(self.mpc.model)
            sim.settings.t_step = self.mpc.settings.t_step
            m = self.mpc.model                              # plant parameters = the controller's: stage 0 / first scenario
            if m.n_tvp:
                tv = sim.get_tvp_template()

                def tvp_fun(t):
                    tv.master[:] = np.asarray(self.mpc.tvp_fun(t).master, float).reshape(-1)[:m.n_tvp]
                    return tv
                sim.set_tvp_fun(tvp_fun)
            if m.n_p:
                pt = sim.get_p_template()

                def p_fun(t):
                    pt.master[:] = np.asarray(self.mpc.p_fun(t).master, float).reshape(-1)[:m.n_p]
                    return pt
                sim.set_p_fun(p_fun)
            sim.setup()
            self.simulator = sim
        self.flags["setup"] = True

    # ---- files ---------------------------------------------------------------------------------------------------
    def _dirs(self):
        name = self._settings.dataset_name
        base = _os.path.join(self._settings.data_dir, name)
        return name, base, _os.path.join(base, "samples_" + name)

    def approx_mpc_sampling_plan_box(self):
        """(x0, u_prev) uniform in the box of the bounds, exported like _ampc_sampler.py:234-273."""
        assert self.flags["setup"], "Sampler was not setup yet. Please call Sampler.setup()."
        st = self._settings
        name, base, _ = self._dirs()
        sp = SamplingPlanner(overwrite=st.overwrite_sampler, id_precision=int(np.ceil(np.log10(st.n_samples))))
        sp.data_dir = base + _os.sep
        sp.set_sampling_var("x0", lambda: np.random.uniform(st.lbx, st.ubx))
        sp.set_sampling_var("u_prev", lambda: np.random.uniform(st.lbu, st.ubu))
        sp.gen_sampling_plan(n_samples=st.n_samples)
        sp.export("sampling_plan_" + name)

    def _plan(self):
        name, base, _ = self._dirs()
        with open(_os.path.join(base, "sampling_plan_" + name + ".pkl"), "rb") as f:
            return _pickle.load(f)

    def _run(self, batch_function, post):
        import pandas as pd
        st = self._settings
        name, base, samples = self._dirs()
        plan = self._plan()
        sampler = Sampler(plan, overwrite=st.overwrite_sampler, sample_name="sample", print_progress=False)
        sampler.data_dir = samples + _os.sep
        sampler.set_batch_function(batch_function)
        sampler.sample_data()
        dh = DataHandler(plan, sample_name="sample")
        dh.data_dir = samples + _os.sep
        for key, fun in post.items():
            dh.set_post_processing(key, fun)
        pd.DataFrame(dh[:]).to_pickle(_os.path.join(base, "data_" + name + "_all.pkl"))
        pd.DataFrame(dh.filter(output_filter=lambda status: status == True)).to_pickle(   # noqa: E712
            _os.path.join(base, "data_" + name + "_opt.pkl"))

    def _chunks(self, n):
        c = self._settings.chunk or int(getattr(self.mpc.settings, "max_batch", 1) or 1)
        return [(lo, min(n, lo + c)) for lo in range(0, n, c)]

    def approx_mpc_open_loop_sampling(self):
        """_ampc_sampler.py:275-360: per row (u0, stats) with stats = {t_make_step, success, iter_count, t_wall_total}."""
        mpc = self.mpc

        def solve_rows(x0, u_prev):
            x0, u_prev = x0.reshape(len(x0), -1), u_prev.reshape(len(u_prev), -1)
            out = []
            for lo, hi in self._chunks(len(x0)):
                r = open_loop_samples(mpc, {"x0": x0[lo:hi], "u_prev": u_prev[lo:hi]})
                out += [(r["u0"][i].reshape(-1, 1), {"t_make_step": float(r["t_make_step"][i]), "success": bool(r["status"][i]),
                                                    "iter_count": int(r["iter_count"][i]), "t_wall_total": float(r["t_wall"][i])})
                        for i in range(hi - lo)]
            return out

        self._run(solve_rows, {"u0": lambda x: x[0], "status": lambda x: x[1]["success"],
                               "t_make_step": lambda x: x[1]["t_make_step"], "t_wall": lambda x: x[1]["t_wall_total"],
                               "iter_count": lambda x: x[1]["iter_count"]})

    def approx_mpc_closed_loop_sampling(self):
        """_ampc_sampler.py:362-526: per row the closed-loop trajectory from (x0, u_prev); table columns x0 (states along the
        trajectory), u_prev, u0 and status, one table row per valid step like the reference's post-processing."""
        mpc, T = self.mpc, int(self._settings.trajectory_length)

        def run_rows(x0, u_prev):
            x0, u_prev = x0.reshape(len(x0), -1), u_prev.reshape(len(u_prev), -1)
            out = []
            for lo, hi in self._chunks(len(x0)):
                r = closed_loop_samples(mpc, self.simulator, {"x0": x0[lo:hi], "u_prev": u_prev[lo:hi]}, T)
                share = r["t_loop"] / max(hi - lo, 1)       # wall time of the batched loop, per sample
                out += [{"x": r["x"][i], "u": r["u"][i], "u_prev": r["u_prev"][i], "success": r["success"][i],
                         "n_valid": int(r["n_valid"][i]), "iter_count": r["iter_count"][i], "t": share} for i in range(hi - lo)]
            return out

        # the reference's table (_ampc_sampler.py:497-503): u0 = simulator.data['_u'], x0 = simulator.data['_x'] (one row per
        # executed plant step), u_prev = u_prev_total (trajectory_length rows, zero behind the step that failed), status /
        # iter_count of the last solve, t_make_step / t_wall of the sample; `n_valid` is an extra column
        def executed(x):
            return x["n_valid"]                                 # plant steps that were executed (the failed solve's step is not)

        def u_prev_total(x):
            out = np.zeros_like(x["u_prev"])
            n = min(x["n_valid"] + 1, len(out))                # (assigned before every solve, including the one that failed)
            out[:n] = x["u_prev"][:n]
            return out

        def last(x):
            return min(x["n_valid"], len(x["u"]) - 1)
        self._run(run_rows, {"u0": lambda x: x["u"][:executed(x)], "x0": lambda x: x["x"][:executed(x)], "u_prev": u_prev_total,
                             "status": lambda x: bool(x["n_valid"] == len(x["u"])), "t_make_step": lambda x: x["t"],
                             "t_wall": lambda x: x["t"], "iter_count": lambda x: int(x["iter_count"][last(x)]),
                             "n_valid": lambda x: x["n_valid"]})

    def default_sampling(self):
        assert self.flags["setup"], "MPC was not setup yet. Please call Sampler.setup()."
        self.approx_mpc_sampling_plan_box()
        if self._settings.closed_loop_flag:
            self.approx_mpc_closed_loop_sampling()
        else:
            self.approx_mpc_open_loop_sampling()
