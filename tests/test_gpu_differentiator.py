"""do_mpc_amd.differentiator on the HIP path (twin of test_differentiator.py)."""
import pytest

import differentiator_common as dc
from test_gpu_parity import make_mpc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["batch_reactor", "industrial_poly"])
def test_linear_parameter_sensitivities_match_the_oracles_sparse_kkt_solve(name):
    dc.check_against_oracle_kkt(make_mpc, name)


def test_sensitivities_match_finite_differences_of_complete_resolves():
    dc.check_against_resolves(make_mpc, "batch_reactor", [("_x0", "S_s"), ("_u_prev", "inp"), ("_p", 0, "S_in")])


def test_sensitivities_of_a_model_with_nl_cons_rows_and_soft_constraints():
    dc.check_against_resolves(make_mpc, "CSTR", [("_x0", "C_b"), ("_u_prev", "Q_dot")])


@pytest.mark.parametrize("name", ["batch_reactor", "CSTR"])
def test_batched_newton_directions_with_several_workspace_slots(name):
    dc.check_batched_directions_equal_single_rows(make_mpc, name, max_batch=8)


@pytest.mark.parametrize("name", ["batch_reactor", "CSTR"])
def test_status_object_licq_sc_and_the_constraint_jacobian(name):
    dc.check_status_and_jacobian(make_mpc, name)


@pytest.mark.parametrize("name", ["batch_reactor", "CSTR", "oscillating_masses"])
def test_active_set_reduction_equals_the_references_reduced_kkt_system(name):
    dc.check_active_set_reduction(make_mpc, name)


def test_a_singular_reduced_system_is_reported():
    dc.check_singular_reduced_system_is_reported(make_mpc)
