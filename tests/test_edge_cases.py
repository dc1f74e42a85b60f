"""Edge cases of the hot path on the host emulation of the kernels: shortest horizons, a horizon shorter than the
branching depth allows, empty batches, one-problem batches, and repeated solves on one handle."""
import numpy as np
import pytest

import hostemu
import parity_common as pc
from do_mpc_amd.examples import CASES
from oracle import ipm


def make(name, **kw):
    ex = CASES[name]
    with hostemu.patched():
        return ex.build_mpc(ex.build_model(), **kw)


@pytest.mark.parametrize("name,kw", [("batch_reactor", {"n_horizon": 1}), ("batch_reactor", {"n_horizon": 2}),
                                      ("CSTR", {"n_horizon": 1, "n_robust": 1}), ("CSTR", {"n_horizon": 3, "n_robust": 0}),
                                      ("oscillating_masses", {"n_horizon": 1})])
def test_short_horizons_match_the_oracle(name, kw):
    """One- to three-stage problems are nearly flat in some input directions (the optimum of the CSTR moves by
    3e-3 between mu = 1e-9 and mu = 0, see parity_common.py), so the comparison is on what is determined: the
    optimal cost (1e-7 relative), the KKT conditions evaluated with the oracle's functions, and u0 at 1e-3."""
    ex = CASES[name]
    mpc = make(name, **kw)
    nlp = pc.oracle_nlp(name, **kw)
    assert (nlp.n_opt_x, nlp.n_g) == (mpc.structure.n_opt_x, mpc.structure.n_g)
    mpc.x0 = ex.X0
    mpc.set_initial_guess()
    u0 = mpc.make_step(ex.X0).ravel()
    assert mpc.solver_stats["success"], mpc.solver_stats
    p = nlp.opt_p(ex.X0, np.zeros(nlp.nu))
    r = ipm.solve(nlp, nlp.initial_guess(ex.X0), p)
    assert r["stats"]["success"]
    f_ours, f_ref = nlp.f(mpc.opt_x_num.master, p), nlp.f(r["x"], p)
    assert abs(f_ours - f_ref) <= 1e-7 * max(1.0, abs(f_ref)), (f_ours, f_ref)
    assert pc.relerr(u0, nlp.u0_of(r["x"])) < 1e-3
    pc.check_kkt_with_oracle_functions(mpc, nlp, ex.X0)


def test_n_robust_larger_than_horizon_is_refused_like_the_reference():
    with pytest.raises(Exception, match="n_robust"):
        make("CSTR", n_horizon=2, n_robust=3)


def test_empty_and_single_batches():
    ex = CASES["batch_reactor"]
    mpc = make("batch_reactor", max_batch=4)
    ps = mpc.structure
    r0 = mpc.S.solve_batch(np.zeros((0, ps.n_opt_x)), mpc._lb_opt_x.master, mpc._ub_opt_x.master, mpc._nlp_cons_lb,
                           mpc._nlp_cons_ub, np.zeros((0, ps.n_opt_p)))
    assert r0["x"].shape == (0, ps.n_opt_x) and r0["stats"].shape == (0,)
    r1 = mpc.make_step_batch(ex.X0[None, :])
    mpc.x0 = ex.X0
    mpc.set_initial_guess()
    u = mpc.make_step(ex.X0).ravel()
    assert r1["stats"]["success"].all() and pc.relerr(r1["u0"][0], u) < 1e-12


def test_repeated_solves_on_one_handle_are_reproducible():
    ex = CASES["CSTR"]
    mpc = make("CSTR")
    out = []
    for _ in range(3):
        mpc.x0 = ex.X0
        mpc.u0 = np.zeros(2)
        mpc._t0 = mpc._t0 * 0
        mpc.opt_x_num.master[:] = 0.0      # set_initial_guess fills _x / _u only (_mpc.py:969-971): the slack and unused slots
        mpc.set_initial_guess()            # would carry the previous solution, and they are part of the barrier problem
        out.append((mpc.make_step(ex.X0).ravel().copy(), mpc.opt_x_num.master.copy(), mpc.solver_stats["iter_count"]))
    for u, x, it in out[1:]:
        assert np.array_equal(u, out[0][0]) and np.array_equal(x, out[0][1]) and it == out[0][2]


def _cstr_with_constraints(vector: bool):
    """CSTR with two soft constraints (T_R <= 137 and C_b <= 0.9), posed as one 2x1 expression or as two scalars."""
    from do_mpc_amd import sym
    from do_mpc_amd.controller import MPC
    ex = CASES["CSTR"]
    model = ex.build_model()
    with hostemu.patched():
        mpc = MPC(model)
        st = mpc.settings
        st.n_horizon, st.n_robust, st.t_step = 6, 1, 0.005
        st.collocation_deg, st.collocation_ni = 2, 1
        st.supress_ipopt_output()
        for k, v in (("T_R", 100), ("T_K", 100)):
            mpc.scaling["_x", k] = v
        mpc.scaling["_u", "Q_dot"] = 2000
        mpc.scaling["_u", "F"] = 100
        track = (model.x["C_b"] - 0.6) ** 2
        mpc.set_objective(mterm=track, lterm=track)
        mpc.set_rterm(F=0.1, Q_dot=1e-3)
        mpc.bounds["lower", "_u", "F"] = 5
        mpc.bounds["lower", "_u", "Q_dot"] = -8500
        mpc.bounds["upper", "_u", "F"] = 100
        mpc.bounds["upper", "_u", "Q_dot"] = 0.0
        if vector:
            mpc.set_nl_cons("both", sym.vertcat(model.x["T_R"], model.x["C_b"]), ub=np.array([137.0, 0.9]), soft_constraint=True,
                            penalty_term_cons=np.array([1e2, 3e2]), maximum_violation=np.array([5.0, 0.2]))
        else:
            mpc.set_nl_cons("tr", model.x["T_R"], ub=137.0, soft_constraint=True, penalty_term_cons=1e2, maximum_violation=5.0)
            mpc.set_nl_cons("cb", model.x["C_b"], ub=0.9, soft_constraint=True, penalty_term_cons=3e2, maximum_violation=0.2)
        mpc.set_uncertainty_values(alpha=np.array([1.0, 1.05, 0.95]), beta=np.array([1.0, 1.1, 0.9]))
        mpc.setup()
    mpc.x0 = ex.X0
    mpc.set_initial_guess()
    u0 = mpc.make_step(ex.X0).ravel()
    return u0, mpc.opt_x_num.master.copy(), np.array(mpc.lam_g_num), dict(mpc.solver_stats), mpc.structure


def test_vector_valued_nl_cons_equals_one_constraint_per_row():
    """optimizer.py:483-585: a vector expression contributes one row and one slack element per entry."""
    uv, xv, lv, sv, psv = _cstr_with_constraints(True)
    us, xs, ls, ss, pss = _cstr_with_constraints(False)
    assert (psv.ne, psv.ns) == (pss.ne, pss.ns) == (2, 2) and psv.n_opt_x == pss.n_opt_x and psv.n_g == pss.n_g
    assert sv["success"] and ss["success"] and sv["iter_count"] == ss["iter_count"]
    assert np.array_equal(uv, us) and np.array_equal(xv, xs) and np.array_equal(lv, ls)


def test_stop_request_returns_user_requested_stop():
    """dompc_abort: a raised stop request makes a solve leave its IPM loop at the first check with status 6
    (IPOPT's User_Requested_Stop); lowering it re-arms the handle."""
    ex = CASES["batch_reactor"]
    mpc = make("batch_reactor")
    mpc.x0 = ex.X0
    mpc.set_initial_guess()
    mpc.S.abort(True)
    mpc.make_step(ex.X0)
    assert mpc.solver_stats["return_status"] == "User_Requested_Stop" and mpc.solver_stats["success"] is False
    assert mpc.solver_stats["iter_count"] == 0
    mpc.S.abort(False)
    mpc.u0 = np.zeros(1)
    mpc._t0 = mpc._t0 * 0
    mpc.set_initial_guess()
    mpc.make_step(ex.X0)
    assert mpc.solver_stats["success"], mpc.solver_stats


def test_too_many_unknowns_per_interval_is_refused_by_name():
    """The kernels eliminate at most 64 (single finite element; 128 with several) collocation / algebraic unknowns per control interval (static_assert in
    csrc/dompc_edge.h): the setup says so instead of leaving the user with a failed hipcc run."""
    from do_mpc_amd.controller import MPC
    from do_mpc_amd.model import Model
    model = Model("continuous")
    xs = [model.set_variable("_x", f"x{i}") for i in range(17)]
    u = model.set_variable("_u", "u")
    for i in range(17):
        model.set_rhs(f"x{i}", -xs[i] + u)
    model.setup()
    mpc = MPC(model)
    st = mpc.settings
    st.n_horizon, st.n_robust, st.t_step = 3, 0, 0.1
    st.collocation_deg, st.collocation_ni = 3, 1               # 4 * 17 = 68 unknowns per interval
    mpc.set_objective(mterm=xs[0] ** 2, lterm=xs[0] ** 2)
    mpc.set_rterm(u=0.1)
    with pytest.raises(NotImplementedError, match="68 collocation"):
        mpc.setup()
