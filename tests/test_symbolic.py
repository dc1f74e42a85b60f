"""Expression DAG: AD against finite differences, code emission against direct evaluation, structs."""
import numpy as np
import pytest

from do_mpc_amd import sym as S
from do_mpc_amd.structs import Entry, Layout, NumStruct


def _fd_jac(fn, x, eps=1e-6):
    f0 = fn(x)
    J = np.zeros((f0.size, x.size))
    for j in range(x.size):
        e = np.zeros_like(x)
        e[j] = eps
        J[:, j] = (fn(x + e) - fn(x - e)) / (2 * eps)
    return J


def test_forward_and_reverse_ad_match_finite_differences():
    x = S.SX.sym("x", 4)
    u = S.SX.sym("u", 2)
    A = np.arange(8.0).reshape(2, 4) / 7.0
    y = S.vertcat(A @ x * S.exp(-u[0] * x[1]) + S.sqrt(1.0 + x[2] ** 2), x[0] / (1.0 + u[1] ** 2) + S.sin(x[3]) * S.log(2 + x[0] ** 2))
    f = S.Function("f", [x, u], [y, S.jacobian(y, S.vertcat(x, u)), S.gradient(S.sum1(y * y), x)])
    xv, uv = np.array([0.3, -0.7, 1.1, 0.4]), np.array([0.5, -0.2])
    val, J, g = [o.full() for o in f(xv, uv)]

    def fn(v):
        return f(v[:4], v[4:])[0].full().ravel()

    Jfd = _fd_jac(fn, np.concatenate([xv, uv]))
    assert np.allclose(J, Jfd, atol=1e-8)
    gfd = _fd_jac(lambda v: np.array([np.sum(fn(np.concatenate([v, uv])) ** 2)]), xv)
    assert np.allclose(g.ravel(), gfd.ravel(), atol=1e-7)


def test_hessian_is_symmetric_and_matches_fd():
    x = S.SX.sym("x", 3)
    L = x[0] ** 3 * S.exp(x[1]) + x[1] * x[2] ** 2 / (1 + x[0] ** 2)
    H, g = S.hessian(L, x)
    fH = S.Function("H", [x], [H, g])
    xv = np.array([0.4, -0.3, 0.9])
    Hn = fH(xv)[0].full()
    assert np.allclose(Hn, Hn.T, atol=1e-12)
    Hfd = _fd_jac(lambda v: fH(v)[1].full().ravel(), xv)
    assert np.allclose(Hn, Hfd, atol=1e-6)


def test_numpy_matmul_with_symbols_and_batch_eval():
    x = S.SX.sym("x", (4, 1))
    A = np.eye(4) * 2.0 + 0.1
    y = A @ x + np.array([[1.0], [2.0], [3.0], [4.0]]) @ S.SX.sym("u", (1, 1))
    assert y.shape == (4, 1)
    f = S.Function("f", [x], [S.sum1(x ** 2)])
    out = f.eval(np.arange(8.0).reshape(4, 2))[0]
    assert np.allclose(out, [[0 + 4 + 16 + 36, 1 + 9 + 25 + 49]])


def test_structural_keys_make_codegen_order_independent_of_history():
    a1, b1 = S.symbol("aa"), S.symbol("bb")
    e1 = S.add(S.mul(a1, b1), S.mul(b1, a1))
    _ = [S.symbol(f"junk{i}") for i in range(50)]       # perturb creation order
    b2, a2 = S.symbol("bb"), S.symbol("aa")
    e2 = S.add(S.mul(b2, a2), S.mul(a2, b2))
    c1 = S.emit_c([("out", e1)], {a1.idx: "A", b1.idx: "B"})
    c2 = S.emit_c([("out", e2)], {a2.idx: "A", b2.idx: "B"})
    assert c1 == c2


def test_power_indexing_layout_matches_canonical_order():
    inner = Layout([Entry("a"), Entry("v", (2, 1))])
    lay = Layout([Entry("_x", struct=inner, repeat=[3, 2]), Entry("_u", struct=Layout([Entry("q")]), repeat=[2])])
    st = NumStruct(lay, 0.0)
    assert lay.size == 3 * 2 * 3 + 2
    st["_x", 1, 0, "v"] = [5.0, 6.0]
    assert np.allclose(st.master[(1 * 2 + 0) * 3 + 1:(1 * 2 + 0) * 3 + 3], [5, 6])
    st["_x", 2, :, "a"] = 7.0
    assert st.master[(2 * 2 + 0) * 3] == 7.0 and st.master[(2 * 2 + 1) * 3] == 7.0
    st["_u", 1] = 9.0
    assert st.master[-1] == 9.0
    assert float(st["_x", 1, 0, "v", 1]) == 6.0


def test_numpy_evaluation_covers_every_operator_with_batched_inputs():
    """Function.eval is generated from the operator tables (no C-text translation): inverse trigonometric functions,
    sign and the derivative of fabs must evaluate on arrays (these ran into NameError / TypeError before)."""
    from do_mpc_amd import sym
    x = sym.SX.sym("x", 3)
    f = sym.Function("f", [x], [sym.vertcat(sym.asin(x[0]), sym.acos(x[1]), sym.atan(x[2]), sym.sign(x[0] - 0.5),
                                            sym.fabs(x[1] - 0.45), sym.fmax(x[0], x[1]), sym.fmin(x[0], x[2]))])
    X = np.array([[0.1, 0.6, -0.3], [0.2, 0.3, 0.4], [1.0, -2.0, 3.0]])       # (numel, batch)
    out = f.eval(X)[0]
    ref = np.vstack([np.arcsin(X[0]), np.arccos(X[1]), np.arctan(X[2]), np.sign(X[0] - 0.5), np.abs(X[1] - 0.45),
                     np.maximum(X[0], X[1]), np.minimum(X[0], X[2])])
    assert out.shape == ref.shape and np.allclose(out, ref, rtol=0, atol=1e-15)
    g = sym.Function("g", [x], [sym.jacobian(sym.fabs(x[0] - 0.3) * x[1] + sym.atan(x[2]), x)])       # d|x|/dx = sign(x)
    J = g.eval(X)[0]
    assert J.shape == (3, 3)
    assert np.allclose(J[0], np.sign(X[0] - 0.3) * X[1]) and np.allclose(J[1], np.abs(X[0] - 0.3))
    assert np.allclose(J[2], 1.0 / (1.0 + X[2] ** 2))
    # single-point call path
    assert np.allclose(np.asarray(f(np.array([0.1, 0.2, 1.0])).arr).ravel(), ref[:, 0])


def test_fmin_fmax_derivatives_follow_casadis_rule():
    """casadi/core/calculus.hpp OP_FMIN / OP_FMAX: one for the selected argument, zero for the other, one half each at a tie"""
    x = S.SX.sym("x", 2)
    y = S.vertcat(S.fmin(x[0] ** 2, 3.0 * x[1]), S.fmax(x[0], S.sin(x[1])) * x[0])
    f = S.Function("f", [x], [y, S.jacobian(y, x), S.hessian(S.sum1(y), x)[0]])
    for xv in (np.array([0.5, 0.4]), np.array([2.0, 0.3]), np.array([-0.3, 1.2])):
        val, J, H = [o.full() for o in f(xv)]
        assert np.allclose(val.ravel(), [min(xv[0] ** 2, 3 * xv[1]), max(xv[0], np.sin(xv[1])) * xv[0]])
        assert np.allclose(J, _fd_jac(lambda v: f(v)[0].full().ravel(), xv), atol=1e-7)
        assert np.allclose(H, _fd_jac(lambda v: f(v)[1].full().sum(axis=0), xv), atol=1e-6)
    a = S.SX.sym("a", 1)
    g = S.Function("g", [a], [S.jacobian(S.fmin(a, 2.0 * a), a), S.jacobian(S.fmax(a, 2.0 * a), a)])
    tie_min, tie_max = [float(o.full()) for o in g(np.array([0.0]))]
    assert tie_min == 1.5 and tie_max == 1.5                      # (0.5 * 1 + 0.5 * 2 at the tie a = 2 a = 0)
