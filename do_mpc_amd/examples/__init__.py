"""The four BASELINE workloads and four more of the reference's shipped NMPC examples (bicycle models, kite, rotating masses),
restated on do_mpc_amd's Model/MPC surface.

Each module has `build_model()` and `build_mpc(model, **settings_overrides)` plus `X0`, the
example's initial state.  Equations/settings follow the reference examples (cited per file);
the un-edited reference templates themselves run through do_mpc_amd.casadi_compat in
tests/test_reference_templates.py when /root/reference is present.
"""
from . import (batch_reactor, bicycle, cstr, dip, industrial_poly, kite, oscillating_masses, oscillating_masses_dae,  # noqa: F401
               rotating_masses)

CASES = {"industrial_poly": industrial_poly, "CSTR": cstr, "batch_reactor": batch_reactor,
         "oscillating_masses": oscillating_masses, "kinematic_bicycle": bicycle.kinematic,
         "dynamic_bicycle": bicycle.dynamic, "kite": kite, "rotating_masses": rotating_masses,
         "oscillating_masses_dae": oscillating_masses_dae, "dip": dip}
BASELINE_CASES = ("industrial_poly", "CSTR", "batch_reactor", "oscillating_masses")
