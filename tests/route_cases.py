"""The NLP modifications of the low-level-route tests (prepare_nlp() -> modify -> create_nlp(), /root/reference/do_mpc/optimizer.py:82-215), in one
place: tests/parity_common.py applies them and solves; __graft_entry__.build() applies them to lower and PREBUILD the code objects the GPU
tests of these cases load (a code object belongs to the NLP structure once terms / rows were added).  No oracle import here: every builder
returns the product-side modification and, as plain lambdas over index lists, the same terms for oracle/nlp_extra.py."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Idx:
    """flat indices of the reference's opt_x layout, from the controller's own structure (same numbers as the oracle's nlp.ix / nlp.iu)"""

    def __init__(self, mpc):
        ps = mpc.structure
        self.N, self.nx, self.nu, self.M = ps.N, ps.nx, ps.nu, ps.M
        self.n_scen = list(ps.scenario_tree["n_scenarios"])
        self.ix, self.iu = ps.ix, ps.iu
        self.scaling = np.asarray(mpc.opt_x_scaling.master, float)


def stopped_before_setup(make_mpc, name, **over):
    """the case's controller as the example builds it, stopped before setup() - the entry of the reference's low-level route
    prepare_nlp() -> modify -> create_nlp() (optimizer.py:82-215)"""
    from do_mpc_amd import MPC
    orig = MPC.setup
    MPC.setup = lambda self: None
    try:
        return make_mpc(name, **over)
    finally:
        MPC.setup = orig


def _terminal_docstring(mpc):
    """optimizer.py:91-97, verbatim: `nlp_obj += sum1(vertcat(*opt_x['_x', -1, 0])**2)` - the list holds the collocation states of the last
    interval of scenario 0 and the terminal state (a discrete model: the terminal state only)"""
    nlp = Idx(mpc)
    from do_mpc_amd.sym import sum1, vertcat
    mpc.nlp_obj += sum1(vertcat(*mpc.opt_x["_x", -1, 0]) ** 2)
    i0 = nlp.ix(nlp.N, 0, 0)         # all stored points of the last interval of scenario 0: M collocation states, then the terminal state
    return lambda X, P: sum(X[i0 + a] ** 2 for a in range((nlp.M + 1) * nlp.nx))


def _terms_all_over_the_tree(mpc):
    """terminal cost on the node state of EVERY leaf (one device function, the same expression at all leaves), a stage term that couples
    state and input of one inner node, a term at the root with a parameter (opt_p['_x0']) as weight, and a term in opt_p alone"""
    nlp = Idx(mpc)
    from do_mpc_amd.sym import sum1
    N, nx = nlp.N, nlp.nx
    ks, ss = min(3, N - 1), nlp.n_scen[min(3, N - 1)] - 1
    for s in range(nlp.n_scen[N]):
        mpc.nlp_obj += 0.5 * sum1(mpc.opt_x["_x", N, s, -1] ** 2)
    mpc.nlp_obj += (mpc.opt_x["_u", ks, ss][0] - 0.3 * mpc.opt_x["_x", ks, ss, -1][1]) ** 2 - 0.1 * mpc.opt_x["_x", ks, ss, -1][0]
    mpc.nlp_obj += mpc.opt_p["_x0"][0] * mpc.opt_x["_u", 0, 0][nlp.nu - 1] ** 2 + sum1(mpc.opt_p["_x0"] ** 2)

    def oracle(X, P):
        ex = sum(0.5 * X[nlp.ix(N, s, nlp.M) + a] ** 2 for s in range(nlp.n_scen[N]) for a in range(nx))
        xn, un = nlp.ix(ks, ss, nlp.M), nlp.iu(ks, ss)
        ex += (X[un] - 0.3 * X[xn + 1]) ** 2 - 0.1 * X[xn]
        ex += P[0] * X[nlp.iu(0, 0) + nlp.nu - 1] ** 2 + sum(P[a] ** 2 for a in range(nx))
        return ex
    return oracle


ADDED_COST = {"docstring": _terminal_docstring, "tree": _terms_all_over_the_tree}


def rows_at_three_nodes(mpc, name):
    """inequality rows appended to nlp_cons: a linear state-input row at an inner node that CUTS OFF the reference's stored solution (active
    at the new one), a second row in another slot of the same node, a two-sided nonlinear row at a stage-1 node, and a row at the root
    whose coefficient is a parameter (opt_p['_x0']).  Returns (build for the oracle, lb, ub)."""
    nlp = Idx(mpc)
    N, nu = nlp.N, nlp.nu
    ks, ss = min(3, N - 1), nlp.n_scen[min(3, N - 1)] - 1
    ox, op = mpc.opt_x, mpc.opt_p
    g_ = np.load(os.path.join(GOLD, name + ".npz"))
    gx = g_["mpc._opt_x_num"][0] / nlp.scaling          # the unmodified problem's solution (scaled variables)
    gp = g_["mpc.opt_p_num"][0]
    xn, un, x1, u1, ur = nlp.ix(ks, ss, nlp.M), nlp.iu(ks, ss), nlp.ix(1, 0, nlp.M), nlp.iu(1, 0), nlp.iu(0, 0) + nu - 1
    v1 = gx[un] - 0.3 * gx[xn + 1]
    v2 = gx[x1] ** 2 + gx[u1] ** 2
    v3 = gp[0] * gx[ur]
    blocks = [(vertcat_(ox["_u", ks, ss][0] - 0.3 * ox["_x", ks, ss, -1][1], -ox["_u", ks, ss][0]),
               [-np.inf, -np.inf], [v1 - 0.02 * max(1.0, abs(v1)), 1e3]),
              (ox["_x", 1, 0, -1][0] ** 2 + ox["_u", 1, 0][0] ** 2, [v2 - 10.0 * abs(v2) - 10.0], [v2 + 10.0 * abs(v2) + 10.0]),
              (op["_x0"][0] * ox["_u", 0, 0][nu - 1], [-np.inf], [v3 + 0.05 * max(1.0, abs(v3))])]
    for ex, lb, ub in blocks:
        mpc.nlp_cons.append(ex)
        mpc.nlp_cons_lb.append(np.array(lb))
        mpc.nlp_cons_ub.append(np.array(ub))

    def oracle(X, P):
        return [X[un] - 0.3 * X[xn + 1], -X[un], X[x1] ** 2 + X[u1] ** 2, P[0] * X[ur]]
    return oracle, np.concatenate([b[1] for b in blocks]), np.concatenate([b[2] for b in blocks])


def vertcat_(*a):
    from do_mpc_amd.sym import vertcat
    return vertcat(*a)


