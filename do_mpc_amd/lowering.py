"""Lower the model's expression DAG to device functions for the gfx950 IPM kernels.

This takes the slot of CasADi's AD + VM on the hot path (nlp_f, nlp_g, nlp_grad_f,
nlp_jac_g, nlp_hess_l inside `nlpsol`, /root/reference/do_mpc/controller/_mpc.py:1326-1328)
and of the reference's own (broken) C-codegen route Optimizer.compile_nlp
(/root/reference/do_mpc/optimizer.py:678-729): at MPC.setup() the rhs, stage cost,
terminal cost and nonlinear constraints are differentiated symbolically *per collocation
point / per stage* (never as one big NLP) and emitted as straight-line
`__host__ __device__`-able C++ into one generated header that the kernel source includes.

Everything is expressed in the reference's *scaled* variables
(/root/reference/do_mpc/optimizer.py:804-818, _mpc.py:1152-1157): the generated
`dompc_dyn` returns  h * rhs(x_s*sx, u_s*su, tvp, p) / sx  (h = t_step/ni; 1 for discrete
models), so the collocation rows are  dyn(x_ij) - sum_r C[r,j] x_ir.
"""
from __future__ import annotations

import hashlib
from typing import Dict, List, Sequence

import numpy as np

from . import sym


def _bind(prefix: str, n: int):
    syms = [sym.symbol(f"{prefix}{i}") for i in range(n)]
    binds = {s.idx: f"{prefix}[{i}]" for i, s in enumerate(syms)}
    return syms, binds


def _fmt_array(name: str, vals: Sequence[float], ctype="double") -> str:
    vals = list(np.asarray(vals).reshape(-1))
    if not vals:
        return f"DOMPC_CONST {ctype} {name}[1] = {{0}};"
    body = ", ".join(sym._cfloat(float(v)) if ctype == "double" else str(int(v)) for v in vals)
    return f"DOMPC_CONST {ctype} {name}[{len(vals)}] = {{{body}}};"


def _sym_hessian_upper(L: sym.Node, v: List[sym.Node]):
    """Upper triangle (i<=j) of d2L/dv2 as nodes; mirrored by the caller."""
    g = sym.reverse_gradient(L, v)
    J = sym.forward_jacobian(g, v)
    return g, J


def lower_model(*, nx, nu, np_, ntvp, x_sym, u_sym, tvp_sym, p_sym, rhs, lterm, mterm, nl_exprs,
                nl_slack_index, eps_penalty, sx, su, rterm, h_scale, deg, ni, discrete, C, D,
                name="model", nz=0, z_sym=(), alg=(), sz=(), sp=None, rterm_expr=None, uprev_sym=(), nl_colloc=False,
                arrival=None, xprev_sym=(), lterm_end=False, nl_dup=False, eps_global=False, extras=None, rows=None) -> str:
    """Return the text of the generated header.

    x_sym/u_sym/z_sym/tvp_sym/p_sym: lists of sym.Node (the model's own symbols, unscaled).
    rhs: list[nx] of Node; alg: list[nz] of Node (algebraic equations of a DAE model, optimizer.py:812-813, unscaled rows);
    lterm, mterm: Node; nl_exprs: list[ne] of Node (without the -eps part).
    sp: `_p` scaling - the reference multiplies the parameters by it in the model equations only (optimizer.py:808-812:
    `_p_unscaled = _p * _p_scaling` feeds rhs / alg; cost and nl_cons read opt_p['_p'] as it is, _mpc.py:1230-1275).

    rterm_expr / uprev_sym: user-defined input penalty rterm(x, u, u_prev, z, tvp, p) (_mpc.py:593-677) and the symbols of
    `mpc.u_prev`; None: the default quadratic form with the weights `rterm`.  The reference evaluates it with UNSCALED x, u, z
    and the SCALED previous input (_mpc.py:1263-1269: `opt_p['_u_prev'] / u_scaling` resp. `opt_x['_u', k-1, ...]`).

    arrival / xprev_sym (estimators, _mhe.py:1118-1127): cost of the FREE initial state, an expression in the (unscaled) model
    states and the symbols `xprev_sym` (the previous estimate, handed over in the `_x0` slot of opt_p); lterm_end: the stage
    cost reads the END state of the interval instead of its first one (_mhe.py:1146-1147, 1190-1192: the measurement
    residual of stage k lives at `_x[k+1, -1]`); nl_dup: the nl_cons rows of the last evaluated point are repeated
    (_mhe.py:1186-1188).

    extras (nlp_route.ObjectiveExtras; the route prepare_nlp -> `nlp_obj += ...` -> create_nlp, optimizer.py:82-129): node-local cost
    terms in the canonical symbols extras.cx / cu / cP.  One device function per distinct term, switch-dispatched by the index the
    tables DOMPC_XTRA_LT_ID / DOMPC_XTRA_MT_ID give for an edge; the functions ADD to the edge's stage-cost / terminal-cost record
    (value, gradient, packed Hessian), whose touched entries are forced into the variable part of the compact record.

    rows (nlp_route.ConstraintExtras): inequality rows appended to nlp_cons that stay inside one node.  DOMPC_NE grows by rows.n_slots; the
    generated nl_cons functions return zeros in the extra slots and the switch-dispatched dompc_xrow functions add the row of an edge's
    slot (table DOMPC_XROW_ID[edge * slots + slot], 0 = masked slot).

    Point functions take the stage variables as v = (x (nx), u (nu), z (nz)): for a model without algebraic states
    that is the (x, u) of the optimised kernels, with them the algebraic block is appended (dense DAE path of the kernels).
    """
    ne_base = len(nl_exprs)
    n_xslots = rows.n_slots if rows is not None else 0
    if n_xslots:
        nl_exprs = list(nl_exprs) + [sym.ZERO] * n_xslots
        nl_slack_index = list(nl_slack_index) + [-1] * n_xslots
    ne = len(nl_exprs)
    ns = len(eps_penalty)
    na = nx + nu
    nav = na + nz                    # inputs of a point function
    nf = nx + nz                     # outputs of the dynamics at a point: [h f / sx ; alg]
    sz = np.ones(nz) if (sz is None or len(sz) == 0) else np.asarray(sz, float)
    sp = np.ones(np_) if sp is None else np.asarray(sp, float)
    xs, bx = _bind("xs", nx)
    us, bu = _bind("us", nu)
    zs, bz = _bind("zs", nz)
    ups, bup = _bind("ups", nu)
    tv, bt = _bind("tvp", ntvp)
    pp, bp = _bind("pp", np_)
    lam, bl = _bind("lam", max(nf, ne, 1))
    binds = {}
    for b in (bx, bu, bz, bup, bt, bp, bl):
        binds.update(b)
    # unscaled model symbols -> scaled kernel symbols
    mapping = {}
    for i, s in enumerate(x_sym):
        mapping[s.idx] = sym.mul(xs[i], sym.const(sx[i]))
    for i, s in enumerate(u_sym):
        mapping[s.idx] = sym.mul(us[i], sym.const(su[i]))
    for i, s in enumerate(z_sym):
        mapping[s.idx] = sym.mul(zs[i], sym.const(sz[i]))
    for i, s in enumerate(tvp_sym):
        mapping[s.idx] = tv[i]
    for i, s in enumerate(p_sym):
        mapping[s.idx] = pp[i]
    for i, s in enumerate(uprev_sym):
        mapping[s.idx] = ups[i]          # (scaled, as the reference passes it)
    mapping_dyn = dict(mapping)      # model equations: parameters times their scaling
    for i, s in enumerate(p_sym):
        if sp[i] != 1.0:
            mapping_dyn[s.idx] = sym.mul(pp[i], sym.const(sp[i]))

    def scaled(nodes, mp=None):
        out = sym.substitute_nodes(list(nodes), mapping if mp is None else mp)
        free = [s for s in sym.free_symbols(out) if s.idx not in binds]
        if free:
            raise Exception(f"expression depends on symbols outside (_x,_u,_z,_tvp,_p): {free}")
        return out

    v = xs + us + zs
    f = [sym.mul(sym.div(e, sym.const(sx[i])), sym.const(h_scale)) for i, e in enumerate(scaled(rhs, mapping_dyn))]
    f += list(scaled(alg, mapping_dyn))
    Jf = sym.forward_jacobian(f, v)
    Lf = sym.ZERO
    for i in range(nf):
        Lf = sym.add(Lf, sym.mul(lam[i], f[i]))
    _, Hf = _sym_hessian_upper(Lf, v)

    def emit_fn(sig: str, outs, zero_first=None) -> str:
        body = sym.emit_c(outs, binds, indent="  ")
        return f"DOMPC_FN {sig} {{\n{body}\n}}\n"

    def hess_outs(H, n, name="H"):
        """symmetric matrix, packed upper triangle row by row (index i*n - i*(i-1)/2 + j - i for i <= j;
        symi() in csrc/dompc_kernel.h) - the device writes and re-reads n(n+1)/2 instead of n*n doubles"""
        outs = []
        for i in range(n):
            for j in range(i, n):
                outs.append((f"{name}[{i * n - i * (i - 1) // 2 + j - i}]", H[i][j]))
        return outs

    parts: List[str] = []
    tables: List[str] = []

    def compact(tag: str, sig: str, dense_outs, force=()):
        """Compact twin of a derivative function.  `dense_outs` = [(index inside the function's dense output block, node)]:
        the entries that depend on the inputs are written one after the other into `o` (DOMPC_<tag>_NV values, dense index
        of entry k in DOMPC_<tag>_VIDX[k]); the others are compile-time constants of the model - zeros by structure, weights
        of a quadratic cost - and are listed once (DOMPC_<tag>_CIDX / _CVAL, non-zero ones only).  The kernels keep a dense
        image of the record in LDS, initialised with the constants once per phase, and move only the variable entries
        through HBM (csrc/dompc_kernel.h: MO_COMPACT) - for industrial_poly 82 of the 231 entries of a collocation point."""
        var = [(i, n) for i, n in dense_outs if n.op != "const" or i in force]
        cst = [(i, n.val) for i, n in dense_outs if n.op == "const" and n.val != 0.0 and i not in force]
        body = sym.emit_c([(f"o[{k}]", n) for k, (_, n) in enumerate(var)], binds, indent="  ")
        parts.append(f"DOMPC_FN {sig} {{\n{body}\n}}\n")
        tables.append(f"#define DOMPC_{tag}_NV {len(var)}")
        tables.append(f"#define DOMPC_{tag}_NC {len(cst)}")
        tables.append(_fmt_array(f"DOMPC_{tag}_VIDX", [i for i, _ in var], "int"))
        tables.append(_fmt_array(f"DOMPC_{tag}_CIDX", [i for i, _ in cst], "int"))
        tables.append(_fmt_array(f"DOMPC_{tag}_CVAL", [v for _, v in cst]))
        return {i: k for k, (i, _) in enumerate(var)}

    def packed(H, n):
        return [H[i][j] for i in range(n) for j in range(i, n)]

    sig_dyn_args = "const double* xs, const double* us, const double* zs, const double* tvp, const double* pp"
    outs = [(f"f[{i}]", f[i]) for i in range(nf)]
    parts.append(emit_fn(f"void dompc_dyn_f({sig_dyn_args}, double* f)", outs))
    outs = [(f"f[{i}]", f[i]) for i in range(nf)]
    outs += [(f"J[{i * nav + j}]", Jf[i][j]) for i in range(nf) for j in range(nav)]
    outs += hess_outs(Hf, nav)
    parts.append(emit_fn(f"void dompc_dyn({sig_dyn_args}, const double* lam, double* f, double* J, double* H)", outs))
    # (dense block of a point: f | J row-major | H packed - PT_STRIDE in csrc/dompc_kernel.h)
    compact("DYN", f"void dompc_dyn_c({sig_dyn_args}, const double* lam, double* o)",
            list(enumerate(list(f) + [Jf[i][j] for i in range(nf) for j in range(nav)] + packed(Hf, nav))))

    # node-local cost terms added to nlp_obj: dense output blocks (value | gradient | packed Hessian) in the layout of the record they join
    xtra = {"lt": [], "mt": []}
    xtra_ew = []                    # per function: (value, [(i, d/dw_i)], [(i, j, d2/dw_i dw_j), i <= j]) over the collocation states w of an interval
    xtra_ids = None
    if extras is not None and extras.groups:
        Pq = {j: sym.symbol(f"Pq{j}") for j in extras.cP}
        binds.update({Pq[j].idx: f"P[{j}]" for j in Pq})
        mp_x = {c.idx: xs[i] for i, c in enumerate(extras.cx)}
        mp_x.update({c.idx: us[i] for i, c in enumerate(extras.cu)})
        mp_x.update({c.idx: Pq[j] for j, c in extras.cP.items()})
        ws, bw = _bind("w", len(extras.cw))
        binds.update(bw)
        mp_x.update({c.idx: ws[i] for i, c in enumerate(extras.cw)})
        tabs = extras.tables()
        xtra_ids = {k: tabs[k][1] for k in ("lt", "mt", "ew")}
        for ex in sym.substitute_nodes(tabs["ew"][0], mp_x):
            used = [i for i, w_ in enumerate(ws) if sym.depends_on([ex], [w_])]
            gx, Hx = _sym_hessian_upper(ex, [ws[i] for i in used])
            xtra_ew.append((ex, [(used[a], gx[a]) for a in range(len(used))],
                            [(used[a], used[b], Hx[a][b]) for a in range(len(used)) for b in range(a, len(used))
                             if not (Hx[a][b].op == "const" and Hx[a][b].val == 0.0)]))
        for kind, vv, n_ in (("lt", v, nav), ("mt", xs, nx)):
            for ex in sym.substitute_nodes(tabs[kind][0], mp_x):
                if kind == "mt" and sym.depends_on([ex], us):
                    raise Exception("a cost term at a terminal node depends on an input")
                gx, Hx = _sym_hessian_upper(ex, vv)
                xtra[kind].append([ex] + list(gx[:n_]) + packed(Hx, n_))
    force = {k: {i for blk in xtra[k] for i, n_ in enumerate(blk) if not (n_.op == "const" and n_.val == 0.0)} for k in xtra}

    # stage cost (unweighted; omega applied by the kernel)
    lt = scaled([lterm])[0]
    gl, Hl = _sym_hessian_upper(lt, v)
    parts.append(emit_fn(f"double dompc_lterm_f({sig_dyn_args})", [("double val", lt)]).replace(
        "\n}\n", "\n  return val;\n}\n"))
    outs = [("val[0]", lt)] + [(f"g[{i}]", gl[i]) for i in range(nav)] + hess_outs(Hl, nav)
    parts.append(emit_fn(f"void dompc_lterm({sig_dyn_args}, double* val, double* g, double* H)", outs))
    pos_lt = compact("LT", f"void dompc_lterm_c({sig_dyn_args}, double* o)", list(enumerate([lt] + list(gl[:nav]) + packed(Hl, nav))), force["lt"])

    mt = scaled([mterm])[0]
    if sym.depends_on([mt], us + zs):
        raise Exception("mterm contains invalid symbolic variables as inputs. Must contain only: _x, _tvp, _p")
    gm, Hm = _sym_hessian_upper(mt, xs)
    sig_m = "const double* xs, const double* tvp, const double* pp"
    parts.append(emit_fn(f"double dompc_mterm_f({sig_m})", [("double val", mt)]).replace(
        "\n}\n", "\n  return val;\n}\n"))
    outs = [("val[0]", mt)] + [(f"g[{i}]", gm[i]) for i in range(nx)] + hess_outs(Hm, nx)
    parts.append(emit_fn(f"void dompc_mterm({sig_m}, double* val, double* g, double* H)", outs))
    pos_mt = compact("MT", f"void dompc_mterm_c({sig_m}, double* o)", list(enumerate([mt] + list(gm[:nx]) + packed(Hm, nx))), force["mt"])
    if xtra_ids is not None:
        # switch-dispatched additions: _f value only (trial points of the line search); _c into the compact record; plain into the dense one
        def dispatch(sig, bodies, ret=""):
            cases = "".join(f"    case {q + 1}: {{\n{b}\n      break;\n    }}\n" for q, b in enumerate(bodies))
            return f"DOMPC_FN {sig} {{\n  switch (id) {{\n{cases}    default: break;\n  }}\n{ret}}}\n"
        for kind, pos, n_, args in (("lt", pos_lt, nav, "const double* xs, const double* us, const double* P"),
                                    ("mt", pos_mt, nx, "const double* xs, const double* P")):
            blks = xtra[kind]
            parts.append(dispatch(f"double dompc_xtra_{kind}_f(int id, {args})",
                                  [sym.emit_c([("val", b[0])], binds, indent="      ", accumulate=True) for b in blks],
                                  "  return val;\n").replace(") {\n  switch", ") {\n  double val = 0.0;\n  switch", 1))
            parts.append(dispatch(f"void dompc_xtra_{kind}_c(int id, {args}, double* o)",
                                  [sym.emit_c([(f"o[{pos[i]}]", n2) for i, n2 in enumerate(b) if i in pos], binds, indent="      ", skip_zero=True, accumulate=True)
                                   for b in blks]))
            def dense_lv(i):
                return "val[0]" if i == 0 else (f"g[{i - 1}]" if i <= n_ else f"H[{i - 1 - n_}]")
            parts.append(dispatch(f"void dompc_xtra_{kind}(int id, {args}, double* val, double* g, double* H)",
                                  [sym.emit_c([(dense_lv(i), n2) for i, n2 in enumerate(b)], binds, indent="      ", skip_zero=True, accumulate=True)
                                   for b in blks]))
        if xtra_ew:
            # terms in the collocation states of an interval: value-only function and a function that hands gradient / Hessian entries
            # (index inside w, value) to the caller's accumulators - the dense edge path adds them to its [w | y] blocks (dompc_dae.h)
            args = "const double* w, const double* P"
            parts.append(dispatch(f"double dompc_xtra_ew_f(int id, {args})",
                                  [sym.emit_c([("val", b[0])], binds, indent="      ", accumulate=True) for b in xtra_ew],
                                  "  return val;\n").replace(") {\n  switch", ") {\n  double val = 0.0;\n  switch", 1))
            bodies = []
            for val, gl_, hl_ in xtra_ew:
                outs = [(f"const double g{i}", n2) for i, n2 in gl_] + [(f"const double h{i}_{j}", n2) for i, j, n2 in hl_]
                body = sym.emit_c(outs, binds, indent="      ")
                body += "".join(f"\n      ga({i}, g{i});" for i, _ in gl_) + "".join(f"\n      ha({i}, {j}, h{i}_{j});" for i, j, _ in hl_)
                bodies.append(body)
            parts.append("template <class GA, class HA>\n" + dispatch(f"void dompc_xtra_ew(int id, {args}, GA ga, HA ha)", bodies))
            tables.append("#define DOMPC_XTRA_EW 1   // cost terms in the collocation states of an interval: dense edge path")
            tables.append("#ifndef DOMPC_FORCE_DENSE\n#define DOMPC_FORCE_DENSE 1\n#endif")
            tables.append(_fmt_array("DOMPC_XTRA_EW_ID", xtra_ids["ew"], "int"))
        tables.append("#define DOMPC_XTRA 1      // node-local cost terms added to nlp_obj (dompc_xtra_lt / dompc_xtra_mt, edge -> function tables below)")
        tables.append(_fmt_array("DOMPC_XTRA_LT_ID", xtra_ids["lt"], "int"))
        tables.append(_fmt_array("DOMPC_XTRA_MT_ID", xtra_ids["mt"], "int"))

    # nonlinear constraints (the "- eps" part is linear and handled by the kernel)
    xrow_fns = []
    force_nl = set()
    if n_xslots:
        Pr = {j: sym.symbol(f"Pr{j}") for j in rows.cP}
        binds.update({Pr[j].idx: f"P[{j}]" for j in Pr})
        mp_r = {c.idx: xs[i] for i, c in enumerate(rows.cx)}
        mp_r.update({c.idx: us[i] for i, c in enumerate(rows.cu)})
        mp_r.update({c.idx: Pr[j] for j, c in rows.cP.items()})
        n_edges = rows.mpc.structure.n_edges
        seen = {}
        xrow_pairs = []
        for e, slot, ex, _, _ in rows.rows:
            h = sym.substitute_nodes([ex], mp_r)[0]
            key = (h.idx, slot)
            if key not in seen:
                row = ne_base + slot
                Jh = sym.forward_jacobian([h], v)[0]
                _, Hh = _sym_hessian_upper(sym.mul(lam[row], h), v)
                # dense index inside the NL block (d | Jd row-major | H packed) of every output of this row
                outs = [(row, h)] + [(ne + row * nav + j, Jh[j]) for j in range(nav)]
                k = 0
                for i in range(nav):
                    for j in range(i, nav):
                        outs.append((ne + ne * nav + k, Hh[i][j]))
                        k += 1
                outs = [(i, n_) for i, n_ in outs if not (n_.op == "const" and n_.val == 0.0)]
                xrow_fns.append((h, row, outs))
                seen[key] = len(xrow_fns)
                force_nl.update(i for i, _ in outs)
            xrow_pairs.append((e, slot, seen[key]))
    if ne:
        d = scaled(nl_exprs)
        Jd = sym.forward_jacobian(d, v)
        Ld = sym.ZERO
        for i in range(ne):
            Ld = sym.add(Ld, sym.mul(lam[i], d[i]))
        _, Hd = _sym_hessian_upper(Ld, v)
        outs = [(f"d[{i}]", d[i]) for i in range(ne)]
        parts.append(emit_fn(f"void dompc_nlcons_f({sig_dyn_args}, double* d)", outs))
        outs = [(f"d[{i}]", d[i]) for i in range(ne)]
        outs += [(f"Jd[{i * nav + j}]", Jd[i][j]) for i in range(ne) for j in range(nav)]
        outs += hess_outs(Hd, nav)
        parts.append(emit_fn(f"void dompc_nlcons({sig_dyn_args}, const double* lam, double* d, double* Jd, double* H)", outs))
        pos_nl = compact("NL", f"void dompc_nlcons_c({sig_dyn_args}, const double* lam, double* o)",
                         list(enumerate(list(d) + [Jd[i][j] for i in range(ne) for j in range(nav)] + packed(Hd, nav))), force_nl)
        if n_xslots:
            def dispatch_r(sig, bodies, pre="", ret=""):
                cases = "".join(f"    case {q + 1}: {{\n{b}\n      break;\n    }}\n" for q, b in enumerate(bodies))
                return f"DOMPC_FN {sig} {{\n{pre}  switch (id) {{\n{cases}    default: break;\n  }}\n{ret}}}\n"
            args_r = "const double* xs, const double* us, const double* P"
            parts.append(dispatch_r(f"double dompc_xrow_f(int id, {args_r})",
                                    [sym.emit_c([("val", h)], binds, indent="      ", accumulate=True) for h, _, _ in xrow_fns],
                                    pre="  double val = 0.0;\n", ret="  return val;\n"))
            parts.append(dispatch_r(f"void dompc_xrow_c(int id, {args_r}, const double* lam, double* o)",
                                    [sym.emit_c([(f"o[{pos_nl[i]}]", n_) for i, n_ in outs], binds, indent="      ", accumulate=True)
                                     for _, _, outs in xrow_fns]))

            def dense_lv(i):
                return f"d[{i}]" if i < ne else (f"Jd[{i - ne}]" if i < ne + ne * nav else f"H[{i - ne - ne * nav}]")
            parts.append(dispatch_r(f"void dompc_xrow(int id, {args_r}, const double* lam, double* d, double* Jd, double* H)",
                                    [sym.emit_c([(dense_lv(i), n_) for i, n_ in outs], binds, indent="      ", accumulate=True)
                                     for _, _, outs in xrow_fns]))
            ids = np.zeros(n_edges * n_xslots, np.int32)
            for e, slot, fid in xrow_pairs:
                ids[e * n_xslots + slot] = fid
            tables += ["#define DOMPC_XROW 1         // inequality rows appended to nlp_cons, node-local: extra row slots of the edges (dompc_xrow*)",
                       f"#define DOMPC_XROW_SLOTS {n_xslots}", f"#define DOMPC_XROW_BASE {ne_base}",
                       f"#define DOMPC_XROW_MASKED {n_edges * n_xslots - len(rows.rows)}      // slots without a row: inert, not counted as constraints",
                       _fmt_array("DOMPC_XROW_ID", ids, "int")]
    else:
        parts.append(f"DOMPC_FN void dompc_nlcons_c({sig_dyn_args}, const double* lam, double* o) {{}}\n")
        tables += ["#define DOMPC_NL_NV 0", "#define DOMPC_NL_NC 0", _fmt_array("DOMPC_NL_VIDX", [], "int"),
                   _fmt_array("DOMPC_NL_CIDX", [], "int"), _fmt_array("DOMPC_NL_CVAL", [])]
        parts.append(f"DOMPC_FN void dompc_nlcons_f({sig_dyn_args}, double* d) {{}}\n")
        parts.append(f"DOMPC_FN void dompc_nlcons({sig_dyn_args}, const double* lam, double* d, double* Jd, double* H) {{}}\n")

    # arrival cost of a free initial state: value, gradient, packed Hessian over the (scaled) initial state
    xpv, bxp = _bind("xprev", nx)
    binds.update(bxp)
    sig_a = "const double* xs, const double* xprev, const double* tvp, const double* pp"
    if arrival is not None:
        mp_a = dict(mapping)
        for i, s_ in enumerate(xprev_sym):
            mp_a[s_.idx] = xpv[i]
        at = scaled([arrival], mp_a)[0]
        if sym.depends_on([at], us + zs):
            raise Exception("the arrival cost must depend on the initial state, the previous estimate, _tvp and _p only")
        ga, Ha = _sym_hessian_upper(at, xs)
        parts.append(emit_fn(f"double dompc_aterm_f({sig_a})", [("double val", at)]).replace("\n}\n", "\n  return val;\n}\n"))
        outs = [("val[0]", at)] + [(f"g[{i}]", ga[i]) for i in range(nx)] + hess_outs(Ha, nx)
        parts.append(emit_fn(f"void dompc_aterm({sig_a}, double* val, double* g, double* H)", outs))
    # (no arrival cost: csrc/dompc_kernel.h defines empty stand-ins - headers of controllers stay as they were)

    # user-defined input penalty: value, gradient and Hessian over (x, u, u_prev); z-dependence is not lowered
    if rterm_expr is not None:
        rt = scaled([rterm_expr])[0]
        if sym.depends_on([rt], zs):
            raise NotImplementedError("structured HIP backend: an rterm expression that depends on algebraic states")
        vr = xs + us + ups
        nr = len(vr)
        gr, Hr = _sym_hessian_upper(rt, vr)
        sig_r = "const double* xs, const double* us, const double* ups, const double* tvp, const double* pp"
        parts.append(emit_fn(f"double dompc_rterm_f({sig_r})", [("double val", rt)]).replace("\n}\n", "\n  return val;\n}\n"))
        outs = [("val[0]", rt)] + [(f"g[{i}]", gr[i]) for i in range(nr)] + hess_outs(Hr, nr)
        parts.append(emit_fn(f"void dompc_rterm({sig_r}, double* val, double* g, double* H)", outs))
        compact("RT", f"void dompc_rterm_c({sig_r}, double* o)", list(enumerate([rt] + list(gr[:nr]) + packed(Hr, nr))))
    else:
        sig_r = "const double* xs, const double* us, const double* ups, const double* tvp, const double* pp"
        parts.append(f"DOMPC_FN double dompc_rterm_f({sig_r}) {{ return 0.0; }}\n")
        parts.append(f"DOMPC_FN void dompc_rterm({sig_r}, double* val, double* g, double* H) {{}}\n")
        parts.append(f"DOMPC_FN void dompc_rterm_c({sig_r}, double* o) {{}}\n")
        tables += ["#define DOMPC_RT_NV 0", "#define DOMPC_RT_NC 0", _fmt_array("DOMPC_RT_VIDX", [], "int"),
                   _fmt_array("DOMPC_RT_CIDX", [], "int"), _fmt_array("DOMPC_RT_CVAL", [])]

    M = 0 if discrete else (deg + 1) * ni
    hdr = [
        "// GENERATED by do_mpc_amd/lowering.py - do not edit.",
        "#pragma once",
        "#include <math.h>",
        f"#define DOMPC_MODEL_NAME \"{name}\"",
        f"#define DOMPC_NX {nx}", f"#define DOMPC_NU {nu}", f"#define DOMPC_NP {np_}",
        f"#define DOMPC_NTVP {ntvp}", f"#define DOMPC_NE {ne}", f"#define DOMPC_NS {ns}", f"#define DOMPC_NZ {nz}",
        f"#define DOMPC_RTERM_CUSTOM {1 if rterm_expr is not None else 0}",
        *(["#define DOMPC_NL_COLLOC 1      // nl_cons rows at every stored point of the interval (DOMPC_NE rows per point)"] if nl_colloc and ne and not discrete else []),
        *(["#define DOMPC_FREE_ROOT 1      // estimator: free initial state with the arrival cost dompc_aterm (no initial-condition rows)"] if arrival is not None else []),
        *(["#define DOMPC_LT_END 1         // estimator: the stage cost reads the END state of the interval"] if lterm_end else []),
        *(["#define DOMPC_NL_DUP 1         // estimator: the nl_cons rows of the last evaluated point once more"] if nl_dup and ne else []),
        *(["#define DOMPC_EPS_GLOBAL 1     // nl_cons_single_slack: the slack variables are shared by all stages (Schur complement in the solver)"] if eps_global and ne and ns else []),
        f"#define DOMPC_DEG {deg if not discrete else 0}", f"#define DOMPC_NI {ni if not discrete else 1}",
        f"#define DOMPC_M {M}", f"#define DOMPC_DISCRETE {1 if discrete else 0}",
        _fmt_array("DOMPC_C", np.asarray(C).reshape(-1) if not discrete else [0.0]),
        _fmt_array("DOMPC_D", D if not discrete else [0.0]),
        _fmt_array("DOMPC_SX", sx),
        _fmt_array("DOMPC_SU", su),
        _fmt_array("DOMPC_RTERM", rterm),
        _fmt_array("DOMPC_EPS_PEN", eps_penalty),
        _fmt_array("DOMPC_NL_SLACK", nl_slack_index, "int"),
        "",
    ]
    text = "\n".join(hdr) + "\n" + "\n".join(tables) + "\n\n" + "\n".join(parts)
    digest = hashlib.sha256(text.encode()).hexdigest()[:16]
    return text + f"\n#define DOMPC_MODEL_HASH \"{digest}\"\n"


def lower_plant(*, x_sym, u_sym, tvp_sym, p_sym, w_sym, v_sym, rhs, meas, discrete, name="plant", z_sym=(), alg=()) -> str:
    """Header for the batched plant integrator (csrc/dompc_plant.hip): the model's right-hand side and measurement
    function in PHYSICAL units, as the reference's Simulator integrates them
    (/root/reference/do_mpc/simulator.py:363-416: `_rhs_fun(x, u, z, tvp, p, w)`, `_meas_fun` at :822).
    Models with algebraic states (semi-explicit index-1 DAE, simulator.py:381-416 IDAS / :363-378 the root-finding problem of a
    discrete DAE): `plant_alg` returns the algebraic equations and their Jacobian w.r.t. z - the kernel solves them for z by
    Newton's method inside every right-hand-side evaluation."""
    nz = len(z_sym)
    groups = [("x", x_sym), ("u", u_sym), ("tvp", tvp_sym), ("p", p_sym), ("w", w_sym), ("v", v_sym), ("z", z_sym)]
    binds: Dict[int, str] = {}
    for cname, syms in groups:
        for i, s in enumerate(syms):
            binds[s.idx] = f"{cname}[{i}]"
    for what, nodes in (("rhs", rhs), ("meas", meas), ("alg", alg)):
        free = [s for s in sym.free_symbols(list(nodes)) if s.idx not in binds]
        if free:
            raise Exception(f"{what} depends on symbols outside (_x,_u,_z,_tvp,_p,_w,_v): {free}")
    sig = "const double* x, const double* u, const double* tvp, const double* p"
    zarg = ", const double* z" if nz else ""
    parts = []
    body = sym.emit_c([(f"f[{i}]", e) for i, e in enumerate(rhs)], binds, indent="  ")
    parts.append(f"DOMPC_FN void plant_rhs({sig}, const double* w{zarg}, double* f) {{\n{body}\n}}\n")
    body = sym.emit_c([(f"y[{i}]", e) for i, e in enumerate(meas)], binds, indent="  ")
    parts.append(f"DOMPC_FN void plant_meas({sig}, const double* v{zarg}, double* y) {{\n{body}\n}}\n")
    if nz:
        Jz = sym.forward_jacobian(list(alg), list(z_sym))
        outs = [(f"a[{i}]", e) for i, e in enumerate(alg)] + [(f"Jz[{i * nz + j}]", Jz[i][j]) for i in range(nz) for j in range(nz)]
        body = sym.emit_c(outs, binds, indent="  ")
        parts.append(f"DOMPC_FN void plant_alg({sig}, const double* w, const double* z, double* a, double* Jz) {{\n{body}\n}}\n")
    if not discrete:
        # Jacobians for the implicit method of the integrator (SDIRK, dompc_plant.hip): d rhs / d x (row-major nx x nx) and, for
        # models with algebraic states, d rhs / d z (nx x nz) and d alg / d x (nz x nx) - the kernel forms the Jacobian of the
        # reduced ODE  f_x - f_z alg_z^-1 alg_x  from them
        nx = len(x_sym)
        Fx = sym.forward_jacobian(list(rhs), list(x_sym))
        outs = [(f"fx[{i * nx + j}]", Fx[i][j]) for i in range(nx) for j in range(nx)]
        extra = ""
        if nz:
            Fz = sym.forward_jacobian(list(rhs), list(z_sym))
            Ax = sym.forward_jacobian(list(alg), list(x_sym))
            outs += [(f"fz[{i * nz + j}]", Fz[i][j]) for i in range(nx) for j in range(nz)]
            outs += [(f"ax[{i * nx + j}]", Ax[i][j]) for i in range(nz) for j in range(nx)]
            extra = ", double* fz, double* ax"
        body = sym.emit_c(outs, binds, indent="  ")
        parts.append(f"DOMPC_FN void plant_jac({sig}, const double* w{zarg}, double* fx{extra}) {{\n{body}\n}}\n")
    hdr = ["// GENERATED by do_mpc_amd/lowering.py:lower_plant - do not edit.", "#pragma once", "#include <math.h>",
           f"#define PLANT_MODEL_NAME \"{name}\"",
           f"#define PLANT_NX {len(x_sym)}", f"#define PLANT_NU {len(u_sym)}", f"#define PLANT_NP {len(p_sym)}",
           f"#define PLANT_NTVP {len(tvp_sym)}", f"#define PLANT_NW {len(w_sym)}", f"#define PLANT_NV {len(v_sym)}",
           f"#define PLANT_NY {len(meas)}", f"#define PLANT_DISCRETE {1 if discrete else 0}",
           *([f"#define PLANT_NZ {nz}"] if nz else []), ""]
    text = "\n".join(hdr) + "\n" + "\n".join(parts)
    digest = hashlib.sha256(text.encode()).hexdigest()[:16]
    return text + f"\n#define PLANT_MODEL_HASH \"{digest}\"\n"
