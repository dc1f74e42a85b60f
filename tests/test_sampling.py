"""do_mpc_amd.sampling: the approximate-MPC open-loop sampler as one batched solve (host emulation of the kernels)."""
import numpy as np

import hostemu
from do_mpc_amd import sampling
from do_mpc_amd.examples import CASES


def test_open_loop_samples_equal_single_make_steps():
    ex = CASES["batch_reactor"]
    with hostemu.patched():
        mpc = ex.build_mpc(ex.build_model(), max_batch=6)
        one = ex.build_mpc(ex.build_model())
    plan = sampling.sampling_plan_box(ex.X0 * 0.9, ex.X0 * 1.1, [0.0], [0.02], n_samples=6, seed=3)
    assert plan["x0"].shape == (6, 4) and np.all(plan["x0"] >= ex.X0 * 0.9) and np.all(plan["x0"] <= ex.X0 * 1.1)
    res = sampling.open_loop_samples(mpc, plan, chunk=4)
    assert res["status"].all() and res["u0"].shape == (6, 1)
    for i in range(6):      # _ampc_sampler.py:291-297, per sample
        one.reset_history()
        one.x0 = plan["x0"][i]
        one.u0 = plan["u_prev"][i]
        one.set_initial_guess()
        u0 = one.make_step(plan["x0"][i]).ravel()
        assert np.allclose(u0, res["u0"][i], rtol=1e-9, atol=1e-12)
        assert one.solver_stats["iter_count"] == res["iter_count"][i]
    df = sampling.to_dataframe(res)
    assert list(df.columns) == ["id", "x0", "u_prev", "u0", "status", "iter_count", "t_wall", "t_make_step"] and len(df) == 6


# ----------------------------------------------------------------------------------------------------------------------
# The reference's sampling tool chain (plan -> samples on disk -> table), pinned by its own golden table.
import json
import os
import pickle

import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sampling_test_fun.json")


def _tool_chain(tmp_path, batched):
    """The steps of examples/tools/sampling/regular/test_fun/sampling_test.py:10-49 (testing/test_sampling_tools.py)."""
    np.random.seed(123)
    d = str(tmp_path) + os.sep
    sp = sampling.SamplingPlanner()
    sp.set_param(overwrite=True)
    sp.data_dir = d
    sp.set_sampling_var("alpha", lambda: np.random.randn())
    sp.set_sampling_var("beta", lambda: np.random.randint(0, 5))
    sp.gen_sampling_plan(n_samples=10)
    sp.add_sampling_case(alpha=10)
    sp.add_sampling_case(beta=10)
    plan = sp.add_sampling_case(alpha=2, beta=2)
    sampler = sampling.Sampler(plan, overwrite=True, print_progress=False)
    sampler.data_dir = d
    calls = []
    if batched:
        def whole_plan(alpha, beta):
            calls.append(len(alpha))
            return list((alpha * beta).tolist())
        sampler.set_batch_function(whole_plan)
    else:
        def one_row(alpha, beta):
            calls.append(1)
            return alpha * beta
        sampler.set_sample_function(one_row)
    sampler.sample_data()
    dh = sampling.DataHandler(plan)
    dh.data_dir = d
    dh.set_post_processing("res_1", lambda x: x)
    dh.set_post_processing("res_2", lambda x: x ** 2)
    return dh[:], dh.filter(input_filter=lambda alpha: alpha < 0), dh.filter(output_filter=lambda res_1: res_1 < 0), calls, plan


@pytest.mark.parametrize("batched", [False, True])
def test_sampling_tool_chain_reproduces_the_reference_table(tmp_path, batched):
    res, res1, res2, calls, plan = _tool_chain(tmp_path, batched)
    ref = json.load(open(GOLDEN))
    assert res == ref["res"] and res1 == ref["res1"] and res2 == ref["res2"]      # equality, like the reference's test
    assert calls == ([13] if batched else [1] * 13)
    assert sorted(os.listdir(tmp_path)) == ["sample_%03d.pkl" % i for i in range(13)]
    assert pickle.load(open(os.path.join(tmp_path, "sample_012.pkl"), "rb")) == 4


def test_sampler_skips_existing_samples_and_handler_reports_missing_ones(tmp_path):
    d = str(tmp_path) + os.sep
    sp = sampling.SamplingPlanner(id_precision=2)
    sp.set_sampling_var("a")
    sp.set_sampling_var("b", lambda: 7)
    with pytest.raises(AssertionError):
        sp.add_sampling_case(b=1)                       # `a` has no generating function
    with pytest.raises(Exception, match="not a valid sampling variable"):
        sp.add_sampling_case(c=1)
    plan = sp.product(a=[1, 2, 3], b=[10, 20])
    assert [r["id"] for r in plan] == ["%02d" % i for i in range(6)] and plan[3] == {"a": 2, "b": 20, "id": "03"}
    sp.data_dir = d
    sp.export("plan.pkl"); sp.export("plan")
    assert sorted(f for f in os.listdir(d) if f.startswith("plan")) == ["plan.pkl", "plan1.pkl"]
    assert pickle.load(open(d + "plan1.pkl", "rb")) == plan

    seen = []
    s = sampling.Sampler(plan[:4], print_progress=False, sample_name="run")
    s.data_dir = d
    with pytest.raises(AssertionError):
        s.set_sample_function(lambda a, zeta: 0)        # unknown argument
    with pytest.raises(AssertionError):
        s.sample_idx(0)                                 # no sample function yet
    s.set_sample_function(lambda a, b: seen.append((a, b)) or a + b)
    s.sample_idx(2)
    s.sample_data()
    assert seen == [(2, 10), (1, 10), (1, 20), (2, 20)] and s.completion_list == ["02", "00", "01", "03"]
    s.sample_data()
    assert len(seen) == 4                               # files exist, overwrite is off

    dh = sampling.DataHandler(plan, sample_name="run")
    dh.data_dir = d
    assert dh[1] == [{"a": 1, "b": 20, "id": "01", "res": 21}]           # no post-processing: raw result under 'res'
    assert dh[5][0]["res"] is None                                      # not sampled yet
    dh.set_post_processing("double", lambda r: 2 * r)
    dh.set_post_processing("with_row", lambda row, r: r - row["a"])
    tab = dh[[0, 5]]
    assert tab[0] == {"a": 1, "b": 10, "id": "00", "double": 22, "with_row": 10}
    assert tab[1] == {"a": 3, "b": 20, "id": "05", "double": None, "with_row": None}
    assert [r["id"] for r in dh.filter(input_filter=lambda a, b: a == 2 and b == 10)] == ["02"]
    assert dh.pre_loaded_data["id"][:2] == ["01", "05"]


def test_sampler_batch_function_runs_the_whole_plan_as_one_solver_batch(tmp_path):
    """The reference's approximate-MPC sampler is a `Sampler` whose sample function calls `mpc.make_step` per row
    (_ampc_sampler.py:283-320); with `set_batch_function` the same plan is one `make_step_batch` launch."""
    ex = CASES["batch_reactor"]
    with hostemu.patched():
        mpc = ex.build_mpc(ex.build_model(), max_batch=5)
    rng = np.random.default_rng(0)
    sp = sampling.SamplingPlanner()
    sp.set_sampling_var("x0", lambda: ex.X0 * rng.uniform(0.9, 1.1, size=4))
    sp.set_sampling_var("u_prev", lambda: rng.uniform(0.0, 0.02, size=1))
    plan = sp.gen_sampling_plan(5)
    launches = []

    def solve_rows(x0, u_prev):
        r = sampling.open_loop_samples(mpc, {"x0": x0, "u_prev": u_prev})
        launches.append(len(x0))
        return [{"u0": r["u0"][i], "status": bool(r["status"][i]), "iter_count": int(r["iter_count"][i])} for i in range(len(x0))]

    s = sampling.Sampler(plan, print_progress=False)
    s.data_dir = str(tmp_path) + os.sep
    s.set_batch_function(solve_rows)
    s.sample_data()
    assert launches == [5]
    dh = sampling.DataHandler(plan)
    dh.data_dir = s.data_dir
    dh.set_post_processing("u0", lambda r: r["u0"])
    dh.set_post_processing("ok", lambda r: r["status"])
    tab = dh[:]
    direct = sampling.open_loop_samples(mpc, {"x0": np.array([r["x0"] for r in plan]), "u_prev": np.array([r["u_prev"] for r in plan])})
    assert all(t["ok"] for t in tab) and np.array_equal(np.array([t["u0"] for t in tab]), direct["u0"])


def test_ampc_sampler_writes_the_reference_files_from_batched_solves(tmp_path, monkeypatch):
    """do_mpc.approximateMPC.AMPCSampler surface (_ampc_sampler.py:41-526): settings, setup(), default_sampling() and the
    files it leaves behind; the rows equal single make_step calls (open loop) / the batched closed loop."""
    import pandas as pd
    import simulator_common as sc
    ex = CASES["batch_reactor"]
    with hostemu.patched():
        mpc = ex.build_mpc(ex.build_model(), max_batch=4)
        one = ex.build_mpc(ex.build_model())
    for nm, lo, hi in (("X_s", 0.5, 2.0), ("S_s", 0.2, 1.0), ("P_s", 0.0, 1.0), ("V_s", 100.0, 140.0)):
        mpc.bounds["lower", "_x", nm], mpc.bounds["upper", "_x", nm] = lo, hi
    s = sampling.AMPCSampler(mpc)
    s.settings.n_samples, s.settings.dataset_name, s.settings.data_dir = 6, "unit", str(tmp_path)
    np.random.seed(5)
    s.setup()
    s.default_sampling()
    base = os.path.join(tmp_path, "unit")
    assert sorted(os.listdir(base)) == ["data_unit_all.pkl", "data_unit_opt.pkl", "samples_unit", "sampling_plan_unit.pkl"]
    assert sorted(os.listdir(os.path.join(base, "samples_unit"))) == ["sample_%d.pkl" % i for i in range(6)]
    df = pd.read_pickle(os.path.join(base, "data_unit_all.pkl"))
    assert list(df.columns) == ["x0", "u_prev", "id", "u0", "status", "t_make_step", "t_wall", "iter_count"] and len(df) == 6
    assert len(pd.read_pickle(os.path.join(base, "data_unit_opt.pkl"))) == int(df["status"].sum())
    for i in (0, 3, 5):
        one.reset_history()
        one.x0, one.u0 = df["x0"][i], df["u_prev"][i]
        one.set_initial_guess()
        u0 = one.make_step(df["x0"][i])
        assert bool(one.solver_stats["success"]) == bool(df["status"][i])
        if df["status"][i]:
            assert np.allclose(u0, df["u0"][i], rtol=1e-9, atol=1e-12) and one.solver_stats["iter_count"] == df["iter_count"][i]

    # closed loop: tables and files (the batched loop itself needs the GPU - `closed_loop_samples`, test_gpu_simulator.py;
    # here it is replaced by per-sample loops of the single-sample controller and plant on the host emulation)
    sim = sc.make_simulator("batch_reactor", hostemu=True)

    def per_sample_loops(mpc_, simulator, plan, T, device=0):
        n = len(plan["x0"])
        x, u, up = np.zeros((n, T + 1, 4)), np.zeros((n, T, 1)), np.zeros((n, T, 1))
        for i in range(n):
            one.reset_history()
            one.x0, one.u0 = plan["x0"][i], plan["u_prev"][i]
            simulator.x0 = plan["x0"][i]
            one.set_initial_guess()
            x[i, 0], cur = plan["x0"][i], plan["u_prev"][i]
            for k in range(T):
                up[i, k] = cur
                u0 = one.make_step(x[i, k])
                x[i, k + 1] = np.asarray(simulator.make_step(u0)).ravel()
                u[i, k], cur = u0.ravel(), u0.ravel()
        return {"x": x, "u": u, "u_prev": up, "success": np.ones((n, T), bool), "n_valid": np.full(n, T),
                "iter_count": np.full((n, T), 7), "t_loop": 0.5}

    monkeypatch.setattr(sampling, "closed_loop_samples", per_sample_loops)
    c = sampling.AMPCSampler(mpc, simulator=sim)
    c.settings.n_samples, c.settings.dataset_name, c.settings.data_dir = 3, "cl", str(tmp_path)
    c.settings.closed_loop_flag, c.settings.trajectory_length = True, 2
    c.setup()
    c.default_sampling()
    dfc = pd.read_pickle(os.path.join(tmp_path, "cl", "data_cl_all.pkl"))
    # the reference's column names (_ampc_sampler.py:497-503): its approximate-MPC trainer reads x0 / u_prev / u0
    assert set(dfc.columns) >= {"id", "x0", "u0", "u_prev", "status", "t_make_step", "t_wall", "iter_count", "n_valid"}
    assert len(dfc) == 3 and dfc["x0"][0].shape == (2, 4) and dfc["u0"][0].shape == (2, 1) and dfc["u_prev"][0].shape == (2, 1)
    assert dfc["status"].all() and (dfc["iter_count"] == 7).all() and (dfc["t_make_step"] > 0).all()
    assert np.array_equal(dfc["u_prev"][1][1], dfc["u0"][1][0])       # the previous input of step 1 = the input of step 0
    plan = pd.read_pickle(os.path.join(tmp_path, "cl", "sampling_plan_cl.pkl"))
    assert np.array_equal(dfc["x0"][2][0], np.asarray(plan[2]["x0"]).ravel())     # trajectory starts at the plan's x0
