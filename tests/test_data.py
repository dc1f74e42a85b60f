"""Per-step records of the controller (do_mpc.data.MPCData surface kept by do_mpc_amd.controller.MPCData)."""
import numpy as np

import parity_common as pc
from test_hostemu_parity import make_mpc


def test_prediction_queries_follow_the_scenario_tree():
    """data.prediction(('_x', name)) etc. (/root/reference/do_mpc/data.py:246-374): [n_size][stages][leaves], the stage-k
    entry of leaf j read from the scenario slot of its ancestor (`structure_scenario`)."""
    mpc = make_mpc("industrial_poly")
    g = pc.golden("industrial_poly")
    x0 = g["mpc._x"][0]
    mpc.x0 = x0
    mpc.set_initial_guess()
    mpc.make_step(x0)
    N, S = mpc.settings.n_horizon, 9
    sc = np.asarray(mpc.data.meta_data["structure_scenario"])
    assert np.array_equal(sc, g["mpc.meta.structure_scenario"])
    px = mpc.data.prediction(("_x", "T_R"))
    pu = mpc.data.prediction(("_u", "m_dot_f"), t_ind=0)
    pa = mpc.data.prediction(("_aux", "T_adiab")) if "T_adiab" in mpc.model.aux.keys() else None
    assert px.shape == (1, N + 1, S) and pu.shape == (1, N, S)
    X, U = mpc.opt_x_num_unscaled, mpc.opt_x_num_unscaled
    for k in (0, 1, 7, N):
        for j in (0, 4, 8):
            assert px[0, k, j] == np.asarray(X["_x", k, int(sc[k, j]), -1, "T_R"]).ravel()[0]
            if k < N:
                assert pu[0, k, j] == np.asarray(U["_u", k, int(sc[k, j]), "m_dot_f"]).ravel()[0]
    assert np.all(px[0, 0, :] == x0[list(mpc.model.x.keys()).index("T_R")])        # every leaf starts at the measured state
    assert np.all(pu[0, 0, :] == mpc.data["_u"][0, 0])                               # ... and shares the first input
    if pa is not None:
        assert pa.shape == (1, N, S)
    # the golden solution through the same query: first input of every leaf = the stored u0
    assert np.allclose(pu[0, 0, 0], g["mpc._u"][0, 0], rtol=1e-8)


def test_prediction_of_vector_valued_and_time_varying_entries():
    mpc = make_mpc("rotating_masses")
    x0 = np.zeros(8)
    mpc.x0 = x0
    mpc.set_initial_guess()
    mpc.make_step(x0)
    mpc.make_step(x0)
    N = mpc.settings.n_horizon
    assert mpc.data.prediction(("_x", "dphi")).shape == (3, N + 1, 1)
    assert mpc.data.prediction(("_x", "dphi", 1)).shape == (1, N + 1, 1)
    assert mpc.data.prediction(("_u", "phi_m_set"), t_ind=0).shape == (2, N, 1)
    tv = mpc.data.prediction(("_tvp", "phi_2_set"))
    assert tv.shape == (1, N + 1, 1)
    assert np.array_equal(tv[0, :, 0], np.asarray(mpc.opt_p_num["_tvp", :, "phi_2_set"]).ravel())
    assert not np.array_equal(mpc.data.prediction(("_x", "phi_2"), t_ind=0), mpc.data.prediction(("_x", "phi_2"), t_ind=1))


def test_results_are_saved_and_loaded_like_in_the_examples(tmp_path):
    """do_mpc.data.save_results / load_results (data.py:376-460) and a pickled model (the reference's test_pickle_unpickle):
    the last lines of every examples/*/main.py."""
    import pickle
    import simulator_common as sc
    from do_mpc_amd import data
    from do_mpc_amd.examples import CASES
    ex = CASES["batch_reactor"]
    model = pickle.loads(pickle.dumps(ex.build_model()))                 # model -> file -> model, then the controller from it
    import hostemu
    with hostemu.patched():
        mpc = ex.build_mpc(model)
    sim = sc.make_simulator("batch_reactor", hostemu=True, model=model)
    g = pc.golden("batch_reactor")
    x0 = g["mpc._x"][0]
    mpc.x0, sim.x0 = x0, x0
    mpc.set_initial_guess()
    for k in range(2):
        u0 = mpc.make_step(x0)
        x0 = sim.make_step(u0)
    assert pc.relerr(mpc.data["_u"], g["mpc._u"][:2]) < 1e-6
    d = str(tmp_path) + "/"
    data.save_results([mpc, sim], "run", d)
    data.save_results([mpc, sim], "run", d)
    data.save_results([mpc], "run", d, overwrite=True)
    import os
    assert sorted(os.listdir(d)) == ["001_run.pkl", "run.pkl"]
    res = data.load_results(d + "001_run.pkl")
    assert set(res) == {"mpc", "simulator"}
    assert np.array_equal(res["mpc"]["_u"], mpc.data["_u"]) and np.array_equal(res["simulator"]["_x"], sim.data["_x"])
    assert np.array_equal(res["mpc"].prediction(("_x", "X_s")), mpc.data.prediction(("_x", "X_s")))
    assert res["mpc"]["_x", "S_s"].shape == (2, 1) and set(data.load_results(d + "run.pkl")) == {"mpc"}
