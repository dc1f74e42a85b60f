"""The reference's shipped template_model.py / template_mpc.py run UN-EDITED on this backend.

Needs /root/reference (present in the build container, absent on the GPU box -> skipped there).
The files are imported from where they lie; nothing is copied.  Each template must give the same structure, bounds, scalings and model function values as the
in-repo restatement in do_mpc_amd/examples/, and on the host emulation the un-edited template reproduces the golden u0.
"""
import importlib.util
import os
import sys

import numpy as np
import pytest

import hostemu
from do_mpc_amd import casadi_compat
from do_mpc_amd.examples import CASES

REF = "/root/reference/examples"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not available")

DIRS = {"industrial_poly": "industrial_poly", "CSTR": "CSTR", "batch_reactor": "batch_reactor",
        "oscillating_masses": "oscillating_masses_discrete"}


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture()
def compat():
    names = casadi_compat.install()
    yield
    casadi_compat.uninstall(names)


@pytest.mark.parametrize("name", list(DIRS))
def test_unedited_templates_lower_to_the_same_model(name, compat):
    d = os.path.join(REF, DIRS[name])
    tm = _load(os.path.join(d, "template_model.py"), f"ref_{name}_template_model")
    tc = _load(os.path.join(d, "template_mpc.py"), f"ref_{name}_template_mpc")
    with hostemu.patched():
        ref_model = tm.template_model()
        ref_mpc = tc.template_mpc(ref_model, silence_solver=True)
        ours = CASES[name].build_mpc(CASES[name].build_model())
    assert ref_mpc.structure.n_opt_x == ours.structure.n_opt_x
    assert ref_mpc.structure.n_g == ours.structure.n_g
    assert np.array_equal(ref_mpc._lb_opt_x.master, ours._lb_opt_x.master)
    assert np.array_equal(ref_mpc._ub_opt_x.master, ours._ub_opt_x.master)
    assert np.array_equal(ref_mpc.opt_x_scaling.master, ours.opt_x_scaling.master)
    # same model functions (the in-repo restatement factors the expressions differently, so the
    # generated text may differ; the values must not)
    rng = np.random.default_rng(0)
    m1, m2 = ref_model, ours.model
    for _ in range(5):
        x = ours._x0.master * 0 + CASES[name].X0 * (1 + 0.01 * rng.standard_normal(m1.n_x))
        u = 0.5 * (ours._u_lb.master + ours._u_ub.master) * (1 + 0.01 * rng.standard_normal(m1.n_u))
        p = ours.p_fun(0.0).master[:m1.n_p]
        z = np.zeros(0)
        args = (x, u, z, np.zeros(m1.n_tvp), p, np.zeros(m1.n_w))
        r1 = m1._rhs_fun.eval(*args)[0]
        r2 = m2._rhs_fun.eval(*args)[0]
        assert np.allclose(r1, r2, rtol=1e-12, atol=1e-12)
    assert "DOMPC_NX %d" % m1.n_x in ref_mpc.generated_header


def test_unedited_industrial_poly_template_reproduces_golden_first_step(compat):
    d = os.path.join(REF, "industrial_poly")
    tm = _load(os.path.join(d, "template_model.py"), "ref_ip_template_model2")
    tc = _load(os.path.join(d, "template_mpc.py"), "ref_ip_template_mpc2")
    with hostemu.patched():
        mpc = tc.template_mpc(tm.template_model(), silence_solver=True)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "industrial_poly.npz"))
    # main.py:59-73 sets x0 through the simulator's struct; same numbers here
    mpc.x0 = g["mpc._x"][0]
    mpc.set_initial_guess()
    u0 = mpc.make_step(g["mpc._x"][0]).ravel()
    assert np.max(np.abs(u0 - g["mpc._u"][0]) / np.maximum(1, np.abs(g["mpc._u"][0]))) < 1e-6
