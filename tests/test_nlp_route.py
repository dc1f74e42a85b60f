"""The low-level route prepare_nlp() -> modify nlp_obj / nlp_cons -> create_nlp() (reference: do_mpc/optimizer.py:82-215, 1050-1094,
do_mpc/controller/_mpc.py:83-128, 323-405) on the structured backend: the attributes exist with the reference's addressing, an
unmodified NLP goes through, every modification that leaves the stage structure is refused BY NAME (never dropped)."""
import numpy as np
import pytest

import hostemu
from do_mpc_amd import MPC
from do_mpc_amd.examples import oscillating_masses as osc
from do_mpc_amd.sym import sum1, vertcat
from do_mpc_amd import nlp_route


def _mpc(setup_now=False):
    """the shipped oscillating_masses controller, stopped before setup()"""
    model = osc.build_model()
    orig = MPC.setup
    MPC.setup = lambda self: None
    try:
        mpc = osc.build_mpc(model)
    finally:
        MPC.setup = orig
    if setup_now:
        with hostemu.patched():
            mpc.setup()
    return mpc


def test_attributes_need_prepare_nlp():
    mpc = _mpc()
    for name in ("opt_x", "opt_p", "nlp_obj", "nlp_cons", "nlp_cons_lb", "nlp_cons_ub", "aux_struct", "opt_x_unscaled"):
        with pytest.raises(AssertionError, match="prior to calling"):
            getattr(mpc, name)


def test_symbolic_structs_follow_the_reference_layout():
    mpc = _mpc()
    mpc.prepare_nlp()
    ps, N = mpc.structure, mpc.settings.n_horizon
    assert mpc.opt_x.shape == (ps.n_opt_x, 1) and mpc.opt_p.shape == (ps.n_opt_p, 1)
    # one vector for a fully indexed entry, a list of vectors for sliced repeats (casadi.tools power indexing)
    u0 = mpc.opt_x["_u", 0, 0]
    assert u0.shape == (1, 1)
    xs = mpc.opt_x["_x", -1, 0]
    assert isinstance(xs, list) and len(xs) == 1 + ps.M and xs[0].shape == (4, 1)
    # the symbols sit at the positions the numeric struct uses
    idx = mpc._opt_x_layout.resolve(("_x", N, 0, -1)).reshape(-1)
    assert [mpc.opt_x.index_of[id(n)] for n in mpc.opt_x["_x", N, 0, -1].nodes()] == list(idx)
    assert mpc.opt_p["_x0"].shape == (4, 1) and mpc.opt_p["_u_prev"].shape == (1, 1)
    assert mpc.aux_struct.shape == (mpc.n_opt_aux, 1)
    # opt_x_unscaled = opt_x * scaling, entry by entry
    mpc2 = _mpc()
    mpc2.scaling["_x", "x"] = np.array([2.0, 1.0, 4.0, 1.0])
    mpc2.prepare_nlp()
    e = mpc2.opt_x_unscaled["_x", 1, 0, -1]
    f = nlp_route.sym.Function("f", [mpc2.opt_x.cat], [e])
    v = np.arange(1.0, mpc2.structure.n_opt_x + 1.0)
    i1 = mpc2._opt_x_layout.resolve(("_x", 1, 0, -1)).reshape(-1)
    assert np.allclose(np.ravel(f.eval(v)[0]), v[i1] * np.array([2.0, 1.0, 4.0, 1.0]))
    # constraint blocks: the structured block with its bounds, as lists
    assert isinstance(mpc.nlp_cons, list) and mpc.nlp_cons[0].shape == (ps.n_g, 1)
    assert len(mpc.nlp_cons_lb) == len(mpc.nlp_cons_ub) == 1 and mpc.nlp_cons_lb[0].shape == (ps.n_g,)


def test_unmodified_route_equals_setup():
    a = _mpc(setup_now=True)
    b = _mpc()
    b.prepare_nlp()
    with hostemu.patched():
        b.create_nlp()
    assert b.flags["setup"] and a.model_hash == b.model_hash
    # after create_nlp the bounds are the concatenations (optimizer.py:1306-1308) and can no longer be replaced by lists
    assert isinstance(b.nlp_cons_lb, np.ndarray) and b.nlp_cons_lb.shape == (b.structure.n_g,)
    with pytest.raises(AssertionError):
        b.nlp_obj = 0
    for m in (a, b):
        m.x0 = osc.X0
        m.set_initial_guess()
    ua, ub = a.make_step(osc.X0), b.make_step(osc.X0)
    assert np.array_equal(ua, ub)


def test_the_reference_docstring_examples():
    """optimizer.py:91-97 (terminal cost): accepted and lowered - equality with an oracle solve of the extended NLP is
    test_hostemu_parity / test_gpu_parity::test_cost_terms_added_to_nlp_obj_same_iterates_as_the_oracle; optimizer.py:131-146 (u_0 = u_1):
    couples two nodes, refused by name"""
    mpc = _mpc()
    mpc.prepare_nlp()
    mpc.nlp_obj += sum1(vertcat(*mpc.opt_x["_x", -1, 0]) ** 2)
    with hostemu.patched():
        mpc.create_nlp()
    assert mpc.flags["setup"] and "#define DOMPC_XTRA 1" in mpc.generated_header
    ps = mpc.structure
    ids = [int(v) for v in mpc.generated_header.split("DOMPC_XTRA_MT_ID[%d] = {" % ps.n_edges)[1].split("}")[0].split(",")]
    assert ids == [0] * (ps.n_edges - 1) + [1]          # the terminal-cost record of the last edge (scenario 0 is the only leaf)
    plain = _mpc(setup_now=True)
    for m in (mpc, plain):
        m.x0 = osc.X0
        m.set_initial_guess()
        m.make_step(osc.X0)
    xN = mpc.opt_x_num["_x", -1, 0, -1]
    assert np.sum(np.square(xN)) < np.sum(np.square(plain.opt_x_num["_x", -1, 0, -1]))      # the terminal state is pulled towards 0

    mpc = _mpc()
    mpc.prepare_nlp()
    extra = mpc.opt_x["_u", 0, 0] - mpc.opt_x["_u", 1, 0]
    mpc.nlp_cons.append(extra)
    mpc.nlp_cons_lb.append(np.zeros(extra.shape))
    mpc.nlp_cons_ub.append(np.zeros(extra.shape))
    with hostemu.patched(), pytest.raises(NotImplementedError, match=r"nlp_cons block 0, row 0: it couples 2 nodes .*_u\[0,0\], _u\[1,0\]"):
        mpc.create_nlp()


def test_added_cost_terms_are_grouped_by_node_and_share_device_functions():
    mpc = _mpc()
    mpc.prepare_nlp()
    N = mpc.settings.n_horizon
    for k in range(1, N):
        mpc.nlp_obj += 0.25 * sum1(mpc.opt_x["_x", k, 0, -1] ** 2) + mpc.opt_x["_u", k, 0][0] ** 2      # the same stage term at N - 1 nodes
    mpc.nlp_obj += mpc.opt_x["_u", 2, 0][0] * mpc.opt_x["_x", 2, 0, -1][1]                               # one more at node (2, 0): joins its group
    with hostemu.patched():
        mpc.create_nlp()
    ps = mpc.structure
    ids = [int(v) for v in mpc.generated_header.split("DOMPC_XTRA_LT_ID[%d] = {" % ps.n_edges)[1].split("}")[0].split(",")]
    assert ids[0] == 0 and ids[2] != ids[1] and all(i == ids[1] for i in ids[3:]) and sorted(set(ids)) == [0, 1, 2]
    # what is not node-local is refused by name, addend by addend
    mpc = _mpc()
    mpc.prepare_nlp()
    mpc.nlp_obj += mpc.opt_x["_x", 3, 0, -1][0] ** 2 + mpc.opt_x["_x", 3, 0, -1][0] * mpc.opt_x["_x", 4, 0, -1][0]
    with hostemu.patched(), pytest.raises(NotImplementedError, match=r"nlp_obj term 0: one addend couples 2 nodes .*_x\[3,0,-1\], _x\[4,0,-1\]"):
        mpc.create_nlp()


def test_terms_in_the_collocation_states_of_one_interval_go_to_the_dense_edge_path():
    """the docstring example on a CONTINUOUS model touches the collocation states of the last interval: accepted, the generated header
    switches the model to the dense edge path; what couples them with anything else is refused by name"""
    from do_mpc_amd.examples import CASES
    ex = CASES["CSTR"]

    def stopped():
        orig = MPC.setup
        MPC.setup = lambda self: None
        try:
            return ex.build_mpc(ex.build_model())
        finally:
            MPC.setup = orig
    mpc = stopped()
    mpc.prepare_nlp()
    ps = mpc.structure
    assert ps.M > 0 and len(mpc.opt_x["_x", -1, 0]) == ps.M + 1
    mpc.nlp_obj += sum1(vertcat(*mpc.opt_x["_x", -1, 0]) ** 2)
    with hostemu.patched():
        mpc.create_nlp()
    h = mpc.generated_header
    assert "#define DOMPC_XTRA_EW 1" in h and "#define DOMPC_FORCE_DENSE 1" in h
    ids = [int(v) for v in h.split("DOMPC_XTRA_EW_ID[%d] = {" % ps.n_edges)[1].split("}")[0].split(",")]
    e_last = int(ps.tables["node_in_edge"][int(ps.tables["level_node_start"][ps.N])])            # the edge into leaf (N, 0)
    assert [i for i, v in enumerate(ids) if v] == [e_last]
    mpc = stopped()
    mpc.prepare_nlp()
    mpc.nlp_obj += mpc.opt_x["_x", 3, 0, 0][0] * mpc.opt_x["_x", 3, 0, -1][0]
    with hostemu.patched(), pytest.raises(NotImplementedError, match="couples collocation states of an interval with other variables"):
        mpc.create_nlp()
    mpc = stopped()
    mpc.prepare_nlp()
    mpc.nlp_obj += mpc.opt_x["_x", 0, 0, 0][0] ** 2                  # stage 0 has no interval behind it: unused entries of opt_x
    with hostemu.patched(), pytest.raises(NotImplementedError, match="unused entries of the reference's opt_x"):
        mpc.create_nlp()


def test_other_misuse_is_caught():
    mpc = _mpc()
    mpc.prepare_nlp()
    # a constraint without its bounds
    mpc.nlp_cons.append(mpc.opt_x["_u", 0, 0])
    with hostemu.patched(), pytest.raises(ValueError, match="one entry per constraint block"):
        mpc.create_nlp()
    # bounds assigned as ONE array in place of the list of blocks: the additions are still checked (never skipped by type)
    mpc = _mpc()
    mpc.prepare_nlp()
    mpc.nlp_cons.append(mpc.opt_x["_u", 0, 0])
    mpc.nlp_cons_lb = np.concatenate([mpc.nlp_cons_lb[0], [0.0]])
    mpc.nlp_cons_ub = np.concatenate([mpc.nlp_cons_ub[0], [0.0]])
    with hostemu.patched(), pytest.raises(ValueError, match="must stay lists"):
        mpc.create_nlp()
    mpc = _mpc()
    mpc.prepare_nlp()
    mpc.nlp_cons_lb = np.array(mpc.nlp_cons_lb[0])          # nothing appended: one array is the structured block's own bounds
    with hostemu.patched():
        mpc.create_nlp()
    assert mpc.nlp_cons_lb.shape == (mpc.structure.n_g,)
    # the structured objective cannot be scaled or replaced
    mpc = _mpc()
    mpc.prepare_nlp()
    with pytest.raises(NotImplementedError, match="can only be EXTENDED"):
        mpc.nlp_obj = 2 * mpc.nlp_obj
    mpc.nlp_obj = mpc.opt_x["_u", 0, 0] ** 2
    with hostemu.patched(), pytest.raises(NotImplementedError, match="was replaced"):
        mpc.create_nlp()
    # symbols from somewhere else
    mpc = _mpc()
    mpc.prepare_nlp()
    mpc.nlp_obj += mpc.model.x["x", 0] ** 2
    with hostemu.patched(), pytest.raises(ValueError, match="neither to mpc.opt_x nor to mpc.opt_p"):
        mpc.create_nlp()
    # a node-local addition is classified as such; as an inequality row it is lowered (an extra row slot of the node's first outgoing edge) ...
    mpc = _mpc()
    mpc.prepare_nlp()
    c = nlp_route.classify(mpc, mpc.opt_x["_x", 3, 0, -1][0] * mpc.opt_x["_u", 3, 0] + mpc.opt_p["_x0"][1])
    assert c["nodes"] == [(3, 0)] and not c["interval_unknowns"] and c["opt_p"] == [1]
    mpc.nlp_cons.append(mpc.opt_x["_x", 3, 0, -1][0] ** 2)
    mpc.nlp_cons_lb.append(np.zeros(1))
    mpc.nlp_cons_ub.append(np.ones(1))
    with hostemu.patched():
        mpc.create_nlp()
    ps = mpc.structure
    assert "#define DOMPC_XROW_SLOTS 1" in mpc.generated_header and "#define DOMPC_XROW_MASKED %d" % (ps.n_edges - 1) in mpc.generated_header
    assert mpc.n_opt_lagr == ps.n_g + 1 and mpc.nlp_cons_ub[-1] == 1.0 and mpc.S.row_mapped
    # ... as an EQUALITY row, at a leaf, or over the unknowns of an interval it is refused by name
    for build, msg in ((lambda m: (m.opt_x["_x", 3, 0, -1][0], 0.5, 0.5), "an EQUALITY row"),
                       (lambda m: (m.opt_x["_x", -1, 0, -1][0], 0.0, 1.0), "a row in the state of a leaf"),
                       (lambda m: (m.opt_x["_u", 2, 0][0], -np.inf, np.inf), "without any finite bound")):
        mpc = _mpc()
        mpc.prepare_nlp()
        ex, lo, hi = build(mpc)
        mpc.nlp_cons.append(ex)
        mpc.nlp_cons_lb.append(np.array([lo]))
        mpc.nlp_cons_ub.append(np.array([hi]))
        with hostemu.patched(), pytest.raises(NotImplementedError, match=msg):
            mpc.create_nlp()


def test_a_term_in_the_parameters_only_changes_nothing():
    a = _mpc(setup_now=True)
    b = _mpc()
    b.prepare_nlp()
    b.nlp_obj += sum1(b.opt_p["_x0"] ** 2)
    with hostemu.patched():
        b.create_nlp()
    for m in (a, b):
        m.x0 = osc.X0
        m.set_initial_guess()
    assert np.array_equal(a.make_step(osc.X0), b.make_step(osc.X0))


def test_every_attribute_the_reference_exposes_at_this_boundary_exists():
    """SURVEY.md 8(b) "attributes other code reads (must survive the rewrite)": do_mpc/differentiator/_nlpdifferentiator.py:803-841,
    do_mpc/data.py:246-374, do_mpc/optimizer.py:448-481 read these off the optimizer object."""
    mpc = _mpc(setup_now=True)
    ps = mpc.structure
    for name in ("nlp_obj", "nlp_cons", "nlp_cons_lb", "nlp_cons_ub", "opt_x", "opt_p", "opt_x_scaling", "opt_x_unscaled", "_lb_opt_x",
                 "_ub_opt_x", "lb_opt_x", "ub_opt_x", "opt_x_num", "opt_x_num_unscaled", "opt_g_num", "lam_g_num", "lam_x_num", "opt_p_num",
                 "opt_aux_num", "aux_struct", "n_opt_x", "n_opt_p", "n_opt_aux", "n_opt_lagr", "n_eps", "scenario_tree", "S", "settings",
                 "flags", "data", "solver_stats"):
        assert hasattr(mpc, name), name
    assert (mpc.n_opt_x, mpc.n_opt_p, mpc.n_opt_lagr) == (ps.n_opt_x, ps.n_opt_p, ps.n_g)
    assert mpc.opt_x.shape == (mpc.n_opt_x, 1) and mpc.opt_p.shape == (mpc.n_opt_p, 1) and mpc.aux_struct.shape == (mpc.n_opt_aux, 1)
    assert np.asarray(mpc.nlp_cons_lb).shape == (mpc.n_opt_lagr,) and mpc.nlp_cons.shape == (mpc.n_opt_lagr, 1)
    for key in ("structure_scenario", "n_branches", "n_scenarios", "parent_scenario", "branch_offset"):
        assert key in mpc.scenario_tree, key
