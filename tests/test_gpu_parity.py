"""Parity tests proper: the HIP path on a real MI355X, through the C ABI (libdompc_ipm.so + the
per-model gfx950 code object), against the oracle and the reference's golden vectors."""
import os

import numpy as np
import pytest

import parity_common as pc
from do_mpc_amd.examples import CASES

pytestmark = pytest.mark.gpu


def make_mpc(name, **kw):
    ex = CASES[name]
    return ex.build_mpc(ex.build_model(), **kw)


class DevArr:
    def __init__(self, a):
        import torch
        self.t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()
        self.ptr = self.t.data_ptr()


def _from_dev(d):
    import torch
    torch.cuda.synchronize()
    return d.t.cpu().numpy()


def test_native_code_is_what_runs():
    mpc = make_mpc("oscillating_masses")
    import ctypes
    assert isinstance(mpc.S._lib, ctypes.CDLL) and "libdompc_ipm.so" in mpc.S._lib._name
    assert mpc.S.num_slots >= 1 and mpc.S.workspace_bytes > 0


@pytest.mark.parametrize("name,steps", [("oscillating_masses", 5), ("batch_reactor", 5), ("CSTR", 5), ("industrial_poly", 5),
                                        ("rotating_masses", 5), ("oscillating_masses_dae", 5), ("dip", 2)])
def test_golden_replay(name, steps):
    pc.check_golden_replay(make_mpc, name, steps)


@pytest.mark.parametrize("name", ["oscillating_masses", "batch_reactor", "industrial_poly", "rotating_masses",
                                  "oscillating_masses_dae"])
def test_newton_direction_matches_sparse_kkt_solve(name):
    pc.check_newton_step(make_mpc, name)


@pytest.mark.parametrize("name", ["batch_reactor", "industrial_poly"])
def test_newton_direction_with_inertia_correction(name):
    """delta_w > 0: W'W and W'w0 of every edge are formed on demand from the stored W"""
    pc.check_newton_step(make_mpc, name, delta=0.05)


@pytest.mark.parametrize("name", ["oscillating_masses", "batch_reactor", "industrial_poly"])
def test_sweep_blocks_match_oracle_jacobian(name):
    mpc = make_mpc(name, max_batch=8)
    pc.check_sweep_blocks(mpc, name, DevArr, _from_dev)


def test_sweep_in_the_launch_shape_of_large_batches():
    """dompc_sweep_batch_device for B >= 4096 iterates runs one 64-thread workgroup per iterate (the launch shape of a solve of that many
    problems, the code object built for it): residuals against the oracle's g, and bitwise the same as the small-batch launch
    (256-thread workgroups, general code object)"""
    name = "industrial_poly"
    B, K = 4096, 6
    mpc = make_mpc(name, max_batch=B)
    ps = mpc.structure
    nlp = pc.oracle_nlp(name)
    g = pc.golden(name)
    rng = np.random.default_rng(5)
    s = nlp.scaling_vector()
    Xk = np.stack([g["mpc._opt_x_num"][k % 5] / s * (1 + 1e-3 * rng.standard_normal(ps.n_opt_x)) for k in range(K)])
    Lk = np.stack([g["mpc._lam_g_num"][k % 5] for k in range(K)])
    Pk = np.stack([pc.golden_opt_p(name, g, k % 5, nlp.n_opt_p) for k in range(K)])
    idx = np.arange(B) % K
    dX, dL, dP = DevArr(Xk[idx]), DevArr(Lk[idx]), DevArr(Pk[idx])
    dG = DevArr(np.zeros((B, ps.n_g)))
    mpc.S.sweep_batch_device(B, dX.ptr, dL.ptr, dP.ptr, dG.ptr, 0)
    G = _from_dev(dG)
    for k in range(K):
        gv = nlp.g(Xk[k], Pk[k])
        assert np.max(np.abs(G[k] - gv)) < 1e-10 * max(1.0, np.max(np.abs(gv)))
    assert np.array_equal(G, G[:K][idx])                     # every copy of an iterate: the same bits
    small = make_mpc(name, max_batch=8)
    dG2, dX2, dL2, dP2 = DevArr(np.zeros((K, ps.n_g))), DevArr(Xk), DevArr(Lk), DevArr(Pk)
    small.S.sweep_batch_device(K, dX2.ptr, dL2.ptr, dP2.ptr, dG2.ptr, 0)
    assert np.array_equal(_from_dev(dG2), G[:K])


@pytest.mark.parametrize("name,opts", [("batch_reactor", None), ("rotating_masses", None),
                                       ("industrial_poly", None), ("CSTR", None)])
def test_same_iterates_as_the_oracle(name, opts):
    """IPOPT regularises every iteration of these problems (free unused variables make its matrix singular at delta_w = 0);
    the driver mirrors the delta_w sequence and keeps the bounded unused variables in the barrier problem."""
    mpc = pc.check_same_iterates_as_oracle(make_mpc, name, oracle_opts=opts)
    assert mpc.solver_stats["n_reg"] == mpc.solver_stats["iter_count"]


# continuous models with the docstring's example: the terms in the collocation states send the model through the dense edge path, which has no
# adjoint recovery of the continuity multipliers - same iterations, primal 2e-12, multipliers of the flat direction next to
# active state bounds 1.5e-4 of the largest (1.2 on 3 299); both multiplier vectors are stationary to 1e-9 (asserted in the check)
_XTRA_LAM_TOL = {("industrial_poly", "docstring"): 5e-4}


@pytest.mark.parametrize("name,which", [("oscillating_masses", "docstring"), ("oscillating_masses", "tree"), ("industrial_poly", "tree"),
                                        ("CSTR", "tree"), ("batch_reactor", "tree"), ("oscillating_masses_dae", "tree"),
                                        ("rotating_masses", "tree"), ("CSTR", "docstring"), ("batch_reactor", "docstring"),
                                        ("industrial_poly", "docstring")])
def test_cost_terms_added_to_nlp_obj_same_iterates_as_the_oracle(name, which):
    """optimizer.py:82-129: `nlp_obj += ...` between prepare_nlp() and create_nlp() - node-local terms (the docstring's own example among
    them) lowered into per-node device functions; every edge path (four edges per wavefront, nl_cons rows, several finite elements,
    dense / DAE, discrete) against an oracle solve of the same extended NLP"""
    def create(mpc):
        mpc.settings.max_batch = 4096          # (workspace slots for the batch launch below)
        mpc.create_nlp()
    mpc = pc.check_added_cost_terms(make_mpc, create, name, which, lam_tol=_XTRA_LAM_TOL.get((name, which), 1e-5))
    if name in ("industrial_poly", "CSTR") and which == "tree":
        # the same problem as members of a batch launch (one wavefront per problem: the batch-shape code object of the extended model)
        x0 = pc.golden(name)["mpc._x"][0]
        B = 4096
        r = mpc.make_step_batch(np.tile(x0, (B, 1)))
        assert np.all(r["stats"]["success"]) and np.all(r["stats"]["iter_count"] == mpc.solver_stats["iter_count"])
        assert pc.relerr(r["x"][B // 3], mpc.opt_x_num.master) < 1e-9 and np.array_equal(r["x"][0], r["x"][B - 1])


@pytest.mark.parametrize("name,over", [("batch_reactor", dict(n_horizon=7)), ("CSTR", dict(n_horizon=5, n_robust=0))])
def test_odd_number_of_edges_same_iterates_as_the_oracle(name, over):
    """The forward pass takes two edges per wavefront: with an odd number of edges the second half of the last pair repeats an edge and
    stores nothing (every shipped case has an even number); same iterations and solution as the oracle's solve of the same NLP"""
    mpc = pc.check_same_iterates_as_oracle(make_mpc, name, **over)
    assert mpc.structure.n_edges % 2 == 1


@pytest.mark.parametrize("name,with_cost", [("oscillating_masses", False), ("CSTR", False), ("CSTR", True), ("industrial_poly", False),
                                            ("industrial_poly", True), ("batch_reactor", False), ("rotating_masses", False)])
def test_rows_appended_to_nlp_cons_same_iterates_as_the_oracle(name, with_cost):
    """optimizer.py:131-215: node-local inequality rows appended to nlp_cons between prepare_nlp() and create_nlp() (extra row slots of the
    node's first outgoing edge; g / lam_g come back in the reference's row order), alone and together with added cost terms"""
    pc.check_added_rows(make_mpc, lambda mpc: mpc.create_nlp(), name, with_cost=with_cost)


@pytest.mark.parametrize("name,over", pc.NONCONVEX_CASES)
def test_nonconvex_examples_reach_the_oracles_local_solution(name, over):
    """Second-order correction + inertia correction: same local minimum as the IPOPT-default oracle with exact inertia."""
    mpc = pc.check_against_oracle_solve(make_mpc, name, oracle_opts=dict(inertia="ldl"), **over)
    if name == "kinematic_bicycle":
        assert mpc.solver_stats["n_soc"] >= 1


def test_baseline_config_cstr_nominal_deg3_vs_oracle():
    pc.check_against_oracle_solve(make_mpc, "CSTR", n_robust=0, collocation_deg=3)


def test_baseline_config_batch_reactor_n50_vs_oracle():
    pc.check_against_oracle_solve(make_mpc, "batch_reactor", n_horizon=50)


def test_industrial_poly_variant_b_tree_vs_oracle():
    """BASELINE configs[3] second reading: 3 combinations, n_robust=2 (9 leaves)."""
    mpc = make_mpc("industrial_poly", n_robust=2, uncertainty="paired")
    nlp = pc.oracle_nlp("industrial_poly", n_robust=2, p_values=pc.PAIRED_P)
    ex = CASES["industrial_poly"]
    mpc.x0 = ex.X0
    mpc.set_initial_guess()
    u0 = mpc.make_step(ex.X0).ravel()
    assert mpc.solver_stats["success"]
    from oracle import ipm
    r = ipm.solve(nlp, nlp.initial_guess(ex.X0), nlp.opt_p(ex.X0, np.zeros(3)))
    assert pc.relerr(u0, nlp.u0_of(r["x"])) < pc.U_RTOL
    pc.check_kkt_with_oracle_functions(mpc, nlp, ex.X0)
    # idempotence: a warm re-solve from the solution with the same parameters returns the same u0
    it1 = mpc.solver_stats["iter_count"]
    mpc.u0 = np.zeros(3)
    mpc._t0 = mpc._t0 * 0
    u1 = mpc.make_step(ex.X0).ravel()
    assert mpc.solver_stats["success"] and mpc.solver_stats["iter_count"] <= it1
    assert pc.relerr(u1, u0) < 1e-6


def test_full_size_243_leaf_tree_kkt_properties():
    """BASELINE configs[4]: industrial_poly scenario tree scaled to 3^5 = 243 leaves (218 700 variables,
    160 330 constraints, 4 008 edges) on one GPU.  Too large for an oracle solve in test time, so the
    solution is checked through size-independent properties with the oracle's NLP functions."""
    mpc = make_mpc("industrial_poly", n_robust=5, uncertainty="paired")
    ps = mpc.structure
    assert (ps.S, ps.n_opt_x, ps.n_g, ps.n_edges) == (243, 218700, 160330, 4008)
    nlp = pc.oracle_nlp("industrial_poly", n_robust=5, p_values=pc.PAIRED_P)
    ex = CASES["industrial_poly"]
    mpc.x0 = ex.X0
    mpc.set_initial_guess()
    u0 = mpc.make_step(ex.X0).ravel()
    assert mpc.solver_stats["success"], mpc.solver_stats
    pc.check_kkt_with_oracle_functions(mpc, nlp, ex.X0)
    # non-anticipativity is structural: one u per node; the first input is shared by all 243 scenarios
    assert u0.shape == (3,)


def test_batch_is_deterministic_and_equals_single_solves():
    name = "industrial_poly"
    ex = CASES[name]
    import bench
    B = 12
    X0 = bench.synthetic_x0_batch(B)
    mpc = make_mpc(name, max_batch=B)
    r = mpc.make_step_batch(X0)
    assert r["stats"]["success"].all(), r["stats"]["status"]
    r2 = mpc.make_step_batch(X0)
    assert np.array_equal(r["x"], r2["x"])                  # bitwise reproducible
    Xrep = np.tile(X0[:1], (B, 1))
    r3 = mpc.make_step_batch(Xrep)
    assert np.all(r3["x"] == r3["x"][0])                    # identical inputs -> identical outputs on every slot
    for i in (0, B - 1):
        m1 = make_mpc(name)
        m1.x0 = X0[i]
        m1.set_initial_guess()
        u = m1.make_step(X0[i]).ravel()
        assert pc.relerr(r["u0"][i], u) < 1e-9
    # solutions satisfy the oracle's constraints
    nlp = pc.oracle_nlp(name)
    for i in (0, 5):
        p = nlp.opt_p(X0[i], np.zeros(3))
        gv = nlp.g(r["x"][i], p)
        assert np.max(np.abs(gv)) < 1e-7


def test_batch_larger_than_slot_count_round_robins():
    mpc = make_mpc("batch_reactor", max_batch=4)       # 4 slots
    ex = CASES["batch_reactor"]
    rng = np.random.default_rng(3)
    X0 = ex.X0 * (1 + 0.05 * rng.uniform(-1, 1, size=(37, 4)))
    # more problems than slots: persistent workgroups pull from the work counter
    ps = mpc.structure
    P = np.tile(mpc.opt_p_num.master, (37, 1))
    P[:, :4] = X0
    P[:, ps.p_off_p:ps.p_off_uprev] = mpc.p_fun(0.0).master
    Xi = np.zeros((37, ps.n_opt_x))
    Xi[:, :ps.off_z].reshape(37, -1, 4)[:] = X0[:, None, :]
    r = mpc.S.solve_batch(Xi, mpc._lb_opt_x.master, mpc._ub_opt_x.master, mpc._nlp_cons_lb, mpc._nlp_cons_ub, P)
    assert r["stats"]["success"].all(), r["stats"][r["stats"]["success"] == 0]
    nlp = pc.oracle_nlp("batch_reactor")
    from oracle import ipm
    for i in (0, 36):
        ro = ipm.solve(nlp, nlp.initial_guess(X0[i]), nlp.opt_p(X0[i], np.zeros(1)))
        assert pc.relerr(r["x"][i][ps.iu(0, 0)], nlp.u0_of(ro["x"])[0]) < pc.U_RTOL


def test_infeasible_problem_reports_failure_not_exception():
    mpc = make_mpc("industrial_poly", nlpsol_opts={"ipopt.max_iter": 60})
    ex = CASES["industrial_poly"]
    x0 = ex.X0.copy()
    x0[3] += 25.0
    mpc.x0 = x0
    mpc.set_initial_guess()
    u0 = mpc.make_step(x0)
    assert u0.shape == (3, 1) and mpc.solver_stats["success"] is False


def test_wide_mode_equals_single_workgroup_mode(monkeypatch):
    """Small batches run K workgroups per problem synchronised by a device-scope barrier; the result must
    agree with the one-workgroup-per-problem path (only the reduction grouping differs)."""
    name = "industrial_poly"
    ex = CASES[name]
    res = {}
    for K in ("1", "4", "32"):
        monkeypatch.setenv("DOMPC_WIDE", K)
        mpc = make_mpc(name)
        mpc.x0 = ex.X0
        mpc.set_initial_guess()
        res[K] = (mpc.make_step(ex.X0).ravel(), mpc.solver_stats["iter_count"], mpc.opt_x_num.master.copy())
        assert mpc.solver_stats["success"]
    for K in ("4", "32"):
        assert pc.relerr(res[K][0], res["1"][0]) < 1e-9
        assert abs(res[K][1] - res["1"][1]) <= 1
        used = np.ones(res[K][2].size, bool)
        used[make_mpc(name).structure.tables["dummy_idx"]] = False
        assert pc.relerr(res[K][2][used], res["1"][2][used]) < 1e-7
    monkeypatch.delenv("DOMPC_WIDE")
    # a small batch (wide, 8 workgroups per problem) against single solves
    import bench
    X0 = bench.synthetic_x0_batch(12)
    mpc = make_mpc(name, max_batch=12)
    r = mpc.make_step_batch(X0)
    assert r["stats"]["success"].all()
    m1 = make_mpc(name)
    m1.x0 = X0[7]
    m1.set_initial_guess()
    assert pc.relerr(r["u0"][7], m1.make_step(X0[7]).ravel()) < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("name,kw,cut,native", [("CSTR", {}, 1, False), ("CSTR", {}, 1, True),
                                                ("industrial_poly", {"n_robust": 2, "uncertainty": "paired"}, 2, True),
                                                # BASELINE configs[4]: the 243-leaf tree, 27 sub-trees below cut level 3 (9 cut parents)
                                                ("industrial_poly", {"n_robust": 5, "uncertainty": "paired"}, 3, True)])
def test_tree_sharded_solve_on_one_gpu_matches_the_plain_solve(name, kw, cut, native):
    """the tree-sharding kernel variant with a cut and world = 1.  Every exchange goes through the
    device<->host handshake (pinned request/acknowledge words, host service loop) and an RCCL all-reduce on a
    communicator of size 1 - called natively by the runtime (ncclAllReduce) or through torch.distributed (nccl
    group); the result must be the plain single-GPU solve."""
    import torch.distributed as dist
    from do_mpc_amd.examples import CASES
    ex = CASES[name]

    def solve(shard):
        mpc = ex.build_mpc(ex.build_model(), **kw)
        mpc.x0 = ex.X0
        mpc.set_initial_guess()
        if shard:
            mpc.shard_tree(0, 1, cut_level=cut, native_rccl=native)
        u0 = mpc.make_step(ex.X0).ravel().copy()
        return u0, mpc.opt_x_num.master.copy(), dict(mpc.solver_stats), mpc.structure.tables["dummy_idx"]

    u_ref, x_ref, st_ref, dummy = solve(False)
    created = not native and not dist.is_initialized()
    if created:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        u, x, st, _ = solve(True)
    finally:
        if created:
            dist.destroy_process_group()
    keep = np.ones(x.size, bool)
    keep[dummy] = False
    assert st["success"] and abs(st["iter_count"] - st_ref["iter_count"]) <= 1
    assert np.allclose(u, u_ref, rtol=1e-8, atol=0)
    assert pc.relerr(x[keep], x_ref[keep]) < 1e-6          # (several cut parents: sums are formed in another order)


@pytest.mark.parametrize("name", ["oscillating_masses", "batch_reactor", "CSTR", "industrial_poly"])
def test_closed_loop_reproduces_the_reference_trajectory(name):
    """The reference's closed-loop tests (testing/test_*.py) on the HIP path: make_step -> plant (tests/plant.py)
    for 5 steps against the golden inputs and states."""
    from test_closed_loop import CL_RTOL, run_closed_loop
    wu, wx = run_closed_loop(make_mpc, name)
    assert wu < CL_RTOL and wx < CL_RTOL


def test_code_objects_are_the_ones_the_unedited_templates_lower_to():
    """tests/golden/template_hashes.json holds the model hashes that the reference's UN-EDITED template_model.py /
    template_mpc.py lower to (tools/template_hashes.py, checked against the templates themselves by
    tests/test_reference_templates.py where /root/reference exists).  The in-repo cases lower to the same text, so the golden
    replays / oracle comparisons of this module run exactly the templates' gfx950 code objects - on the real GPU."""
    import json
    import os
    pinned = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "template_hashes.json")))
    assert len(pinned) >= 10
    for name, h in pinned.items():
        if name.endswith("_mhe"):          # (the estimator of the example: the chain problem of do_mpc_amd.estimator.MHE)
            ex = CASES[name[:-4]]
            mpc = ex.build_mhe(ex.build_model())._mpc
        else:
            mpc = make_mpc(name)
        assert mpc.model_hash == h, (name, mpc.model_hash, h)
        assert os.path.basename(os.path.dirname(mpc.S.code_object_path)) == h


def test_user_defined_rterm_written_as_the_default_gives_the_default_solution():
    pc.check_custom_rterm_equal_to_default(make_mpc)


@pytest.mark.parametrize("name", ["oscillating_masses", "CSTR"])
def test_user_defined_rterm_vs_oracle(name):
    pc.check_custom_rterm_vs_oracle(make_mpc, name)


@pytest.mark.parametrize("name", ["CSTR", "industrial_poly"])
def test_pivoting_fallback_of_the_factorisation(name, monkeypatch):
    """The blocked elimination on the matrix cores runs in natural pivot order and hands an edge whose threshold test fails to the
    out-of-line factorisation with partial pivoting (phase_edge_factor_pivot) - never taken on the shipped workloads.  A code object
    whose threshold cannot be met (-DDOMPC_GJ_U=1e9) sends EVERY edge of every iteration through it: same iterations, same solution
    (different elimination order: rounding level)"""
    ex = CASES[name]
    sol = []
    for defs in ("", "DOMPC_GJ_U=1e9"):
        monkeypatch.setenv("DOMPC_DEFS", defs)
        mpc = make_mpc(name)
        mpc.x0 = ex.X0
        mpc.set_initial_guess()
        mpc.make_step(ex.X0)
        assert mpc.solver_stats["success"]
        sol.append((mpc.solver_stats["iter_count"], mpc.opt_x_num.master.copy(), mpc.lam_g_num.copy()))
    assert sol[0][0] == sol[1][0]
    assert pc.relerr(sol[0][1], sol[1][1]) < 1e-9 and pc.relerr(sol[0][2], sol[1][2]) < 1e-7
    assert not np.array_equal(sol[0][1], sol[1][1])          # (a different elimination did the work: not the same bits)


def test_register_elimination_of_several_finite_elements_equals_the_pivoting_one(monkeypatch):
    """batch_reactor (two finite elements per interval, 24 unknowns): the collocation block is eliminated in registers in the natural pivot
    order (round 6); -DDOMPC_REG_GJ=0 builds the in-LDS elimination with partial pivoting it replaces, -DDOMPC_GJ_U=1e9 makes the threshold test
    of the new one fail on every edge (its fallback IS the pivoting one).  Same iterations, same solution to rounding, other bits."""
    ex = CASES["batch_reactor"]
    sol = []
    for defs in ("", "DOMPC_REG_GJ=0", "DOMPC_GJ_U=1e9"):
        monkeypatch.setenv("DOMPC_DEFS", defs)
        mpc = make_mpc("batch_reactor")
        mpc.x0 = ex.X0
        mpc.set_initial_guess()
        mpc.make_step(ex.X0)
        assert mpc.solver_stats["success"]
        sol.append((mpc.solver_stats["iter_count"], mpc.opt_x_num.master.copy(), mpc.lam_g_num.copy()))
    assert sol[0][0] == sol[1][0] == sol[2][0]
    assert pc.relerr(sol[0][1], sol[1][1]) < 1e-9 and pc.relerr(sol[0][2], sol[1][2]) < 1e-7
    assert not np.array_equal(sol[0][1], sol[1][1])          # (a different elimination did the work: not the same bits)
    assert np.array_equal(sol[1][1], sol[2][1])              # (the fallback of the new one is the old one)


@pytest.mark.parametrize("name,over,x0", pc.NL_COLLOC_CASES, ids=[c[0] for c in pc.NL_COLLOC_CASES])
def test_nl_cons_at_collocation_points(name, over, x0):
    pc.check_nl_cons_at_collocation_points(make_mpc, name, over, x0)


@pytest.mark.gpu
@pytest.mark.parametrize("over,x0", [c[1:] for c in pc.SINGLE_SLACK_CASES], ids=[c[0] for c in pc.SINGLE_SLACK_CASES])
def test_single_slack_shared_by_all_stages(over, x0):
    pc.check_single_slack(make_mpc, over, x0)


def test_single_slack_batch_equals_single_solves():
    pc.check_single_slack_batch(make_mpc)


@pytest.mark.parametrize("over,o_over,x0", [c[1:] for c in pc.OPEN_LOOP_CASES], ids=[c[0] for c in pc.OPEN_LOOP_CASES])
def test_open_loop_with_several_scenarios(over, o_over, x0):
    pc.check_open_loop(make_mpc, over, o_over, x0)


@pytest.mark.parametrize("name", ["CSTR", "batch_reactor"])
def test_dense_edge_path_reproduces_the_fast_path(name, monkeypatch):
    """a model without algebraic states through the dense edge path of the DAE models (-DDOMPC_FORCE_DENSE=1, own code
    object): same iterates as the single-element fast path (blocked elimination on the matrix cores)"""
    ex = CASES[name]
    sol = []
    for defs in ("", "DOMPC_FORCE_DENSE=1"):
        monkeypatch.setenv("DOMPC_DEFS", defs)
        mpc = make_mpc(name)
        mpc.x0 = ex.X0
        mpc.set_initial_guess()
        mpc.make_step(ex.X0)
        assert mpc.solver_stats["success"]
        sol.append((mpc.solver_stats["iter_count"], mpc.opt_x_num.master.copy(), mpc.lam_g_num.copy()))
    assert sol[0][0] == sol[1][0]
    assert pc.relerr(sol[0][1], sol[1][1]) < 1e-10 and pc.relerr(sol[0][2], sol[1][2]) < 1e-8


def test_mhe_golden_replay():
    """moving horizon estimation on the HIP path: the reference's estimator run (results_rotatingMasses.pkl) step by step"""
    ex = CASES["rotating_masses"]
    pc.check_mhe_golden_replay(lambda: ex.build_mhe(ex.build_model()))


def test_mhe_batch_of_estimation_problems():
    ex = CASES["rotating_masses"]
    pc.check_mhe_batch(lambda **kw: ex.build_mhe(ex.build_model(), **kw))


def test_mhe_with_process_noise_against_the_oracle():
    ex = CASES["rotating_masses"]
    pc.check_mhe_with_process_noise(lambda: ex.build_mhe_w(ex.build_model(process_noise=True)))


def test_mhe_inputs_measured_without_noise_against_the_oracle():
    ex = CASES["rotating_masses"]
    pc.check_mhe_inputs_measured_without_noise(lambda: ex.build_mhe_w(ex.build_model(process_noise=True, input_meas_noise=False), max_batch=2))


@pytest.mark.parametrize("single_slack", [False, True], ids=["slack_per_stage", "single_slack"])
def test_mhe_soft_constraint_against_the_oracle(single_slack):
    ex = CASES["rotating_masses"]
    pc.check_mhe_soft_constraint(lambda **kw: ex.build_mhe_w(ex.build_model(process_noise=True), **kw), single_slack)


def test_mhe_scaling_of_states_inputs_and_estimated_parameters():
    ex = CASES["rotating_masses"]
    pc.check_mhe_scaling_invariance(lambda **kw: ex.build_mhe_w(ex.build_model(process_noise=True), max_batch=1, **kw))


def test_mhe_for_a_model_with_algebraic_states():
    ex = CASES["rotating_masses"]
    pc.check_mhe_dae_equals_ode(lambda dae: ex.build_mhe_w(ex.build_model(process_noise=True, dae=dae)))
    pc.check_mhe_dae_make_step(lambda dae: ex.build_mhe(ex.build_model(dae=dae)))


def test_discrete_time_mhe_against_the_oracle():
    from do_mpc_amd.examples import oscillating_masses as om
    pc.check_discrete_mhe(lambda: om.build_mhe(om.build_model(estimation=True)))
    pc.check_discrete_mhe_dae_equals_ode(lambda dae: om.build_mhe(om.build_model(estimation=True, dae=dae)))


def _rccl_world_worker(rank, world, port, name, kw, cut, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    try:
        ex = CASES[name]
        mpc = ex.build_mpc(ex.build_model(), gpu_index=rank, **kw)
        mpc.x0 = ex.X0
        mpc.set_initial_guess()
        info = mpc.shard_tree(rank, world, cut_level=cut, native_rccl=True)       # dompc_rccl_unique_id / dompc_rccl_init, world > 1
        u0 = mpc.make_step(ex.X0).ravel().copy()
        q.put((rank, u0, mpc.opt_x_num.master.copy(), dict(mpc.solver_stats), int((info["edge_mask"] == 1).sum())))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_native_rccl_exchange_with_two_ranks():
    """The runtime's own RCCL communicator with world = 2 (one process per GPU; unique id from rank 0 broadcast over the
    caller's group, ncclAllReduce called by the host service loop): the two-rank solve of the 9-leaf tree equals the plain
    one.  Needs two devices - skipped on the 1-GPU boxes of the pool."""
    import socket
    import torch
    import torch.multiprocessing as mp
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (dompc_rccl_init with world = 2)")
    name, kw, cut = "industrial_poly", {"n_robust": 2, "uncertainty": "paired"}, 1
    ex = CASES[name]
    ref = ex.build_mpc(ex.build_model(), **kw)
    ref.x0 = ex.X0
    ref.set_initial_guess()
    u_ref = ref.make_step(ex.X0).ravel().copy()
    x_ref, it_ref = ref.opt_x_num.master.copy(), ref.solver_stats["iter_count"]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rccl_world_worker, args=(r, 2, port, name, kw, cut, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    keep = np.ones(x_ref.size, bool)
    keep[ref.structure.tables["dummy_idx"]] = False
    for rank, u, x, st, n_own in res:
        assert st["success"] and abs(st["iter_count"] - it_ref) <= 2 and n_own > 0
        assert np.allclose(u, u_ref, rtol=1e-7, atol=0)
        assert pc.relerr(x[keep], x_ref[keep]) < 1e-6


@pytest.mark.gpu
def test_bench_gpus_flag_on_this_box():
    """`python bench.py --gpus 2` on a 1-GPU box refuses by name; with two devices it reports n_gpus = 2."""
    import json
    import subprocess
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--batch", "64", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline", "--no-b1", "--no-variant-b", "--no-traffic"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=900)
    if torch.cuda.device_count() < 2:
        assert r.returncode != 0 and "needs 2 devices" in (r.stderr + r.stdout)
        return
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 128 and out["solve"]["converged_all_ranks"] == 128


@pytest.mark.gpu
def test_mid_size_tree_same_iterates_as_the_oracle():
    """27-leaf industrial_poly tree (n_robust = 3, 24 300 variables) on the GPU against an oracle SOLVE: same iteration and
    regularisation counts, final iterate 1e-8 - the step between the 9-leaf fixture and the 243-leaf tree (KKT properties only)."""
    pc.check_tree27_same_iterates_as_oracle(make_mpc)


@pytest.mark.gpu
def test_mid_size_tree_sharded_kernel_variant_against_the_oracle():
    """... and the tree-sharded kernel variant on it (cut level 2: 3 cut parents, 9 sub-trees; world = 1, native RCCL exchanges)."""
    pc.check_tree27_same_iterates_as_oracle(make_mpc, shard=dict(rank=0, world=1, cut_level=2, native_rccl=True))


def _oracle_member(args):
    import warnings
    warnings.filterwarnings("ignore")
    i, x0 = args[:2]
    opts = args[2] if len(args) > 2 else None
    from oracle import ipm as oipm
    nlp = pc.oracle_nlp("industrial_poly")
    r = oipm.solve(nlp, nlp.initial_guess(x0), nlp.opt_p(x0, np.zeros(nlp.nu)), opts=opts)
    if opts:
        return i, nlp.u0_of(r["x"]), int(r["stats"]["iter_count"]), bool(r["stats"]["success"]), r["x"], dict(r["stats"])
    return i, nlp.u0_of(r["x"]), int(r["stats"]["iter_count"]), bool(r["stats"]["success"]), r["x"]


@pytest.mark.parametrize("leaves", [81, 243])
def test_large_trees_equal_the_stored_oracle_solve(leaves):
    """BASELINE configs[4] (243 leaves) and the 81-leaf tree against the oracle's SOLVE (tests/golden/oracle_tree*.npz), on the whole
    chip (KArgs::wide_spread)."""
    pc.check_big_tree_against_stored_oracle_solve(make_mpc, leaves)


def test_243_leaf_tree_sharded_code_path_equals_the_stored_oracle_solve():
    pc.check_big_tree_against_stored_oracle_solve(make_mpc, 243, shard=dict(rank=0, world=1, cut_level=3, native_rccl=True))


@pytest.mark.gpu
def test_members_of_the_timed_batch_equal_oracle_solves():
    """The TIMED launch shape of bench.py (B >= 4096: one 64-lane wavefront per problem, 2048 resident slots, problems pulled from
    a device-wide counter, perturbed x0 of bench.synthetic_x0_batch) tied to the oracle directly: twelve members of THE launch the
    bench times (B = 16 384, its default batch: eight rounds over the 2 048 slots; round 4 sampled a B = 4 096 launch) against
    oracle/ipm.solve of the same x0 - same iteration count, u0 and the full primal solution."""
    import multiprocessing as mp
    import bench
    B = 16384
    X0 = bench.synthetic_x0_batch(B)
    mpc = make_mpc("industrial_poly", max_batch=B)
    assert mpc.S.num_slots >= 1024                     # (one wavefront per problem: 8 slots per CU)
    r = mpc.make_step_batch(X0)
    assert r["stats"]["success"].all()
    # (first / later rounds of the work queue; 2048 and 4095: the two findings of round 4; 100, 5000, 12345: late stops of round 5 before the
    #  adjoint recovery of the continuity multipliers)
    members = [0, 1, 100, 2047, 2048, 4095, 5000, 8191, 9999, 12345, 15000, 16383]
    with mp.get_context("spawn").Pool(len(members)) as pool:
        res = pool.map(_oracle_member, [(i, X0[i]) for i in members])
    used = np.ones(mpc.structure.n_opt_x, bool)
    used[mpc.structure.tables["dummy_idx"]] = False
    for i, u_ref, it_ref, ok, x_ref in res:
        assert ok
        # Same iterates AND the same stop.  Until round 5 the termination test could fire up to four iterations after the oracle's (members
        # 2048: 57 / 55, 100: 62 / 58, 12345: 58 / 56, 5000: 57 / 56): the multiplier steps of the continuity rows, d nu = P dx + p, carried
        # an error Sigma_max * eps ~ 1e-8 into the dual residual of the last iterations.  They now come from the x rows of the Newton system
        # itself on the last levels of the barrier parameter (riccati_forward_t<true>): 12 of 12 members stop in the
        # oracle's iteration, the full primal solution agrees to 4e-8 (before: 1.9e-6).
        assert int(r["stats"]["iter_count"][i]) == it_ref, (i, r["stats"]["iter_count"][i], it_ref)
        assert pc.relerr(r["u0"][i], u_ref) < 1e-10, (i, r["u0"][i], u_ref)
        # (the problem has weakly determined entries - a flat direction along which the iterate still travels 1.7e-3 between a 1e-8 and a
        #  1e-10 stop; measured 1e-13 ... 4.5e-9, member 4095: 4.0e-8)
        assert pc.relerr(r["x"][i][used], x_ref[used]) < 2e-7, i


@pytest.mark.gpu
def test_randomly_drawn_members_of_the_timed_batch_equal_oracle_solves():
    """64 members of the B = 16 384 launch the bench times, drawn with numpy's default_rng ("parity of the timed mode is green on
    twelve hand-picked members, not on a random sample"), against oracle/ipm.solve of the same x0: converged, the same solution, and the
    oracle's iteration count.  A member whose count differs is solved again by the oracle with `inertia = "curvature_then_ldl"`: the
    oracle's default inertia test is a curvature PROXY for the exact count IPOPT gets from MUMPS, and the one way the two solvers were
    found to part (round 6, member 15 803 of this draw: iteration 6, the product accepts delta_w = 1.37e-7, the proxy rejects it and
    escalates to 4.5e-3; exact count at 1.37e-7: 7 210 negative eigenvalues = m, i.e. IPOPT accepts) is a factorisation with the
    correct inertia that the proxy rejects.  With the rejections checked exactly the oracle has to take the product's path: the same
    iteration count, at least one such false rejection on its way - anything else fails the test."""
    import multiprocessing as mp
    import os
    import bench
    B = 16384
    X0 = bench.synthetic_x0_batch(B)
    members = sorted(int(i) for i in np.random.default_rng(2026).choice(B, size=64, replace=False))
    mpc = make_mpc("industrial_poly", max_batch=B)
    r = mpc.make_step_batch(X0)
    assert r["stats"]["success"].all()
    it = r["stats"]["iter_count"]
    n_proc = min(len(members), os.cpu_count() or 8)
    with mp.get_context("spawn").Pool(n_proc) as pool:
        res = pool.map(_oracle_member, [(i, X0[i]) for i in members])
    used = np.ones(mpc.structure.n_opt_x, bool)
    used[mpc.structure.tables["dummy_idx"]] = False
    other = []
    for i, u_ref, it_ref, ok, x_ref in res:
        assert ok, i
        # (measured: u0 1e-15 ... 2e-11, full primal solution 1e-13 ... 5e-9 - also on the member whose path differs: same KKT point)
        assert pc.relerr(r["u0"][i], u_ref) < 1e-9, (i, r["u0"][i], u_ref)
        assert pc.relerr(r["x"][i][used], x_ref[used]) < 2e-7, i
        if int(it[i]) != it_ref:
            other.append((i, int(it[i]), it_ref))
    assert len(other) <= 3, other              # (each costs minutes of exact inertia counts below; measured: 1 of 64)
    if other:
        with mp.get_context("spawn").Pool(len(other)) as pool:
            res2 = pool.map(_oracle_member, [(i, X0[i], {"inertia": "curvature_then_ldl"}) for i, _, _ in other])
        for (i, it_gpu, it_ref), (_, u2, it2, ok2, x2, st2) in zip(other, res2):
            assert ok2 and it2 == it_gpu, (i, it_gpu, it_ref, it2)
            assert st2.get("n_proxy_false_rejections", 0) >= 1, (i, st2)
            assert pc.relerr(r["x"][i][used], x2[used]) < 2e-7, i


def test_whole_chip_placement_gives_the_bits_of_the_one_xcd_placement(monkeypatch):
    """KArgs::wide_spread (round 5): the workgroups of ONE problem on all XCDs (the default for a single problem) against the placement
    of round 4 (K workgroups of one XCD) - another mapping of workgroups to the same work items and another barrier flavour (with L2
    write-back), the same arithmetic in the same order: identical solution, multipliers and iteration count."""
    ex = CASES["industrial_poly"]
    out = []
    for spread, K in (("0", "16"), ("1", "16"), ("1", "24")):
        monkeypatch.setenv("DOMPC_WIDE_SPREAD", spread)
        monkeypatch.setenv("DOMPC_WIDE", K)
        mpc = make_mpc("industrial_poly")
        mpc.x0 = ex.X0
        mpc.set_initial_guess()
        mpc.make_step(ex.X0)
        assert mpc.solver_stats["success"]
        out.append((mpc.opt_x_num.master.copy(), np.array(mpc.lam_g_num), mpc.solver_stats["iter_count"]))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and out[0][2] == out[1][2]
    # another number of workgroups regroups the partial sums of the reductions: same iterations, solution to rounding
    assert out[2][2] == out[0][2] and pc.relerr(out[2][0], out[0][0]) < 1e-9


def test_batch_code_object_is_checked_against_the_general_one(tmp_path):
    """the runtime launches `<name>_batch.hsaco` for one-wavefront batches whenever that file exists; a sibling built from other
    sources / for another model must not be used.  dompc_batch_object_state: 1 for the prebuilt pair, 2 (and a warning) when the sibling
    is the batch object of ANOTHER model class."""
    import shutil
    import warnings
    from do_mpc_amd import build
    from do_mpc_amd.solver import HipIpmSolver
    a = make_mpc("batch_reactor", max_batch=4096)
    assert a.S.batch_object_state == 1
    b = make_mpc("CSTR", max_batch=4096)
    src_general, wrong_sibling = a.S.code_object_path, b.S.code_object_path[:-len(".hsaco")] + "_batch.hsaco"
    d = tmp_path / "m"
    d.mkdir()
    shutil.copy(src_general, d / "dompc_gfx950.hsaco")
    shutil.copy(wrong_sibling, d / "dompc_gfx950_batch.hsaco")
    ctor = dict(a.S._ctor)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        s = HipIpmSolver(ctor.pop("structure"), ctor.pop("header_text"), ctor.pop("model_hash"), _lib_path=build.runtime_library(),
                         _code_object=str(d / "dompc_gfx950.hsaco"), **{k: v for k, v in ctor.items() if k in ("nlpsol_opts", "device", "max_batch", "block_threads")})
    assert s.batch_object_state == 2 and any("not used" in str(x.message) for x in w)
    # ... and the handle still solves batches (with the general object)
    ex = CASES["batch_reactor"]
    r = s.solve_batch(np.tile(a.opt_x_num.master * 0 + 0.0, (2, 1)) + 0.0, a._lb_opt_x.master, a._ub_opt_x.master, a._nlp_cons_lb, a._nlp_cons_ub,
                      np.tile(a.opt_p_num.master, (2, 1)))
    assert r["stats"].shape == (2,)
    s.close()


def test_control_interval_with_80_unknowns_same_iterates_as_the_oracle():
    pc.check_interval_with_more_than_64_unknowns(make_mpc)


def test_watchdog_ends_the_crawl_on_the_full_horizon_kite_problem():
    pc.check_watchdog_on_kite_full_horizon(make_mpc)


def test_watchdog_on_the_straggler_of_the_cold_estimator_batch():
    ex = CASES["rotating_masses"]
    pc.check_watchdog_on_mhe_straggler(lambda: ex.build_mhe(ex.build_model()))


def test_newton_direction_on_the_last_barrier_level_satisfies_the_state_rows():
    """(one problem = 256 threads / several workgroups: the bottom-up edge order of the adjoint recovery over several wavefronts)"""
    pc.check_newton_step_at_late_iterate(make_mpc)
