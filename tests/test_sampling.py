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
