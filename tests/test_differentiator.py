"""do_mpc_amd.differentiator on the TEST-ONLY host emulation of the kernels (the -m gpu twin: test_gpu_differentiator.py)."""
import numpy as np
import pytest

import differentiator_common as dc
from test_hostemu_parity import make_mpc


@pytest.mark.parametrize("name", ["batch_reactor", "oscillating_masses", "industrial_poly"])
def test_linear_parameter_sensitivities_match_the_oracles_sparse_kkt_solve(name):
    dc.check_against_oracle_kkt(make_mpc, name)


def test_sensitivities_match_finite_differences_of_complete_resolves():
    dc.check_against_resolves(make_mpc, "batch_reactor", [("_x0", "S_s"), ("_u_prev", "inp"), ("_p", 0, "S_in")])


def test_sensitivities_of_a_model_with_nl_cons_rows_and_soft_constraints():
    """CSTR (nl_cons row with a slack variable `_eps`, 9 scenarios): du0/dx0 and du0/du_prev against re-solves."""
    dc.check_against_resolves(make_mpc, "CSTR", [("_x0", "C_b"), ("_x0", "T_R"), ("_u_prev", "Q_dot")])


def test_reference_surface():
    """`sens_num["dxdp", indexf[...], indexf[...]]` as in examples/batch_reactor_differentiator/main.py:158-168."""
    from do_mpc_amd.differentiator import DoMPCDifferentiator, indexf
    mpc = dc.solved(make_mpc, "batch_reactor")
    nd = DoMPCDifferentiator(mpc)
    nd.settings.check_LICQ = False
    nd.settings.check_rank = False
    nd.settings.lin_solver = "scipy"
    dx_dp, dlam_dp = nd.differentiate()
    du0dx0 = nd.sens_num["dxdp", indexf["_u", 0, 0], indexf["_x0"]]
    du0dup = nd.sens_num["dxdp", indexf["_u", 0, 0], indexf["_u_prev"]].full()
    assert du0dx0.shape == (1, 4) and du0dup.shape == (1, 1)
    assert np.array_equal(np.asarray(nd.sens_num["dxdp"]), np.asarray(dx_dp))
    # the rterm pulls u0 towards u_prev: 0 < du0/du_prev < 1
    assert 0.0 < du0dup[0, 0] < 1.0



def test_batched_newton_directions_with_several_workspace_slots():
    dc.check_batched_directions_equal_single_rows(make_mpc, "batch_reactor", max_batch=8)


@pytest.mark.parametrize("name", ["oscillating_masses", "batch_reactor", "CSTR"])
def test_status_object_licq_sc_and_the_constraint_jacobian(name):
    dc.check_status_and_jacobian(make_mpc, name)


@pytest.mark.parametrize("name", ["oscillating_masses", "batch_reactor", "CSTR"])
def test_active_set_reduction_equals_the_references_reduced_kkt_system(name):
    dc.check_active_set_reduction(make_mpc, name)


def test_a_singular_reduced_system_is_reported():
    dc.check_singular_reduced_system_is_reported(make_mpc)


def test_standalone_nlp_differentiator():
    dc.check_standalone_nlp_differentiator()


def test_rank_test_of_the_licq_check():
    from do_mpc_amd.differentiator import rows_independent
    import scipy.sparse as sps
    rng = np.random.default_rng(0)
    M = rng.standard_normal((6, 9))
    assert rows_independent(M) and rows_independent(sps.csr_matrix(M))
    M[4] = 2.0 * M[1] - M[3]
    assert not rows_independent(M)
    assert not rows_independent(rng.standard_normal((5, 3)))          # more active constraints than variables
    big = sps.random(2000, 2600, density=2e-3, random_state=1, format="csr") + sps.eye(2000, 2600)
    assert rows_independent(big)
    big = sps.vstack([big, big[7] + big[11]], format="csr")
    assert not rows_independent(big)
