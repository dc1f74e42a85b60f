"""The reference's low-level route `prepare_nlp() -> modify nlp_obj / nlp_cons -> create_nlp()`
(/root/reference/do_mpc/optimizer.py:82-215, 1050-1094; /root/reference/do_mpc/controller/_mpc.py:83-128, 323-405).

In the reference `opt_x`, `opt_p` are casadi.tools symbolic structs and `nlp_obj`, `nlp_cons` symbolic expressions over them that the
user may extend before `create_nlp()` hands everything to nlpsol.  Here the NLP never exists as one flat symbolic expression - it is a
tree of stage blocks that the kernels factorise by a Riccati recursion - so these attributes are

* `opt_x`, `opt_p`, `opt_x_unscaled`, `aux_struct`: symbolic structs over the reference's layouts (same power indexing), created on
  first use (8 100 symbols for the shipped industrial_poly problem);
* `nlp_obj`: a handle on the structured objective that records what is ADDED to it (`mpc.nlp_obj += expr`);
* `nlp_cons`, `nlp_cons_lb`, `nlp_cons_ub`: lists whose first entry is the structured constraint block (n_g rows in the reference's
  order) and to which the user appends expressions / bounds;

and `create_nlp()` classifies every addition (`classify`): what the structured solver can take is lowered, everything else is refused
BY NAME - with the variables and tree nodes that make it non-structured - instead of being silently dropped ("the
structured backend must detect 'NLP no longer stage-structured'").
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np

from . import sym
from .structs import Layout, SymStruct


class OptSymStruct(SymStruct):
    """struct_symSX analogue with casadi's list semantics for sliced repeats: `opt_x['_x', -1, 0]` is the list of the (1 + M) state
    vectors of stage N, scenario 0; `opt_x['_u', 0, 0]` one vector."""

    def __init__(self, layout: Layout, prefix: str):
        self.layout = layout
        self.prefix = prefix
        self.vec = sym.SX([sym.symbol(f"{prefix}_{i}") for i in range(layout.size)], (layout.size, 1))
        self.index_of = {id(n): i for i, n in enumerate(self.vec.data)}

    def __getitem__(self, key):
        idx = self.layout.resolve(key)
        if idx.ndim >= 2:          # repeats that were sliced or left out: one vector per repeat
            flat = idx.reshape(-1, idx.shape[-1])
            return [sym.SX([self.vec.data[i] for i in row], (len(row), 1)) for row in flat]
        idx = idx.reshape(-1)
        return sym.SX([self.vec.data[i] for i in idx], (len(idx), 1))


class StructuredBlock:
    """Stands for the n_g constraint rows (or the objective) the structured lowering produces; has a shape, cannot be taken apart."""

    def __init__(self, what: str, rows: int):
        self.what, self.rows = what, int(rows)

    @property
    def shape(self):
        return (self.rows, 1)

    def __repr__(self):
        return f"<structured {self.what}: {self.rows} row{'s' if self.rows != 1 else ''}, lowered to gfx950 stage blocks>"


class NlpObjective(StructuredBlock):
    """`mpc.nlp_obj`: the structured objective plus the terms added to it.  Only additions are recorded; any other operation would
    need the objective as one expression, which does not exist here."""

    def __init__(self, terms: Sequence = ()):
        super().__init__("objective", 1)
        self.terms: List = list(terms)

    def __add__(self, other):
        return NlpObjective(self.terms + [other])

    __radd__ = __add__

    def __sub__(self, other):
        return NlpObjective(self.terms + [-sym._sx(other)])

    def _refuse(self, *a, **k):
        raise NotImplementedError("structured HIP backend: nlp_obj can only be EXTENDED (nlp_obj += expression over opt_x / opt_p); "
                                  "scaling or replacing the structured objective is not supported - use set_objective()")

    __mul__ = __rmul__ = __truediv__ = __rtruediv__ = __neg__ = __pow__ = __rsub__ = _refuse


def describe_variables(mpc, idx_x: Sequence[int]) -> str:
    """tree nodes / intervals the opt_x entries `idx_x` belong to, for the refusal messages"""
    ps = mpc.structure
    where = set()
    for g in idx_x:
        if g < ps.off_z:
            k, r = divmod(int(g), ps.S * (1 + ps.M) * ps.nx)
            s, r = divmod(r, (1 + ps.M) * ps.nx)
            i = r // ps.nx
            where.add(f"_x[{k},{s},{'-1' if i == ps.M else i}]")
        elif g < ps.off_u:
            k, r = divmod(int(g) - ps.off_z, ps.S * max(ps.M, 1) * ps.nz)
            where.add(f"_z[{k},{r // (max(ps.M, 1) * ps.nz)}]")
        elif g < ps.off_eps:
            k, r = divmod(int(g) - ps.off_u, ps.SU * ps.nu)
            where.add(f"_u[{k},{r // ps.nu}]")
        else:
            where.add("_eps")
    out = sorted(where)
    return ", ".join(out[:6]) + (f", ... ({len(out)} blocks)" if len(out) > 6 else "")


def classify(mpc, expr) -> dict:
    """Which optimisation variables / parameters does an added expression touch, and does it stay inside one node of the tree?
    node-local: only the node state `_x[k, s, -1]` and that node's own input `_u[k, s]` (plus opt_p)."""
    ps = mpc.structure
    ex = sym._sx(expr)
    free = sym.free_symbols(ex.nodes())
    ox, op = mpc.opt_x, mpc.opt_p
    ix = sorted(ox.index_of[id(n)] for n in free if id(n) in ox.index_of)
    ip = sorted(op.index_of[id(n)] for n in free if id(n) in op.index_of)
    foreign = [n for n in free if id(n) not in ox.index_of and id(n) not in op.index_of]
    nodes = set()
    colloc = False
    for g in ix:
        if g < ps.off_z:
            k, r = divmod(g, ps.S * (1 + ps.M) * ps.nx)
            s, r = divmod(r, (1 + ps.M) * ps.nx)
            if r // ps.nx != ps.M:
                colloc = True
            nodes.add((k, s))
        elif g < ps.off_u:
            colloc = True
        elif g < ps.off_eps:
            k, r = divmod(g - ps.off_u, ps.SU * ps.nu)
            nodes.add((k, r // ps.nu))
        else:
            colloc = True
    return {"rows": ex.numel(), "opt_x": ix, "opt_p": ip, "foreign": foreign, "nodes": sorted(nodes), "interval_unknowns": colloc,
            "constant": not ix}


def split_additive(node, scale: float = 1.0):
    """[(factor, addend)] with  node = sum factor * addend:  the sum / difference / negation / constant-multiple structure at the top of an
    expression is taken apart, everything below stays one addend."""
    if node.op == "add":
        return split_additive(node.a, scale) + split_additive(node.b, scale)
    if node.op == "sub":
        return split_additive(node.a, scale) + split_additive(node.b, -scale)
    if node.op == "neg":
        return split_additive(node.a, -scale)
    if node.op == "mul" and node.a.op == "const":
        return split_additive(node.b, scale * node.a.val)
    if node.op == "mul" and node.b.op == "const":
        return split_additive(node.a, scale * node.b.val)
    if node.op == "div" and node.b.op == "const":
        return split_additive(node.a, scale / node.b.val)
    return [(scale, node)]


class ObjectiveExtras:
    """Node-local cost terms added to `nlp_obj` (optimizer.py:82-129), grouped by the tree node they belong to and rewritten in the
    canonical symbols of a device function: a term in the state `_x[k, s, -1]` / the input `_u[k, s]` of ONE node (and opt_p) joins

    * the stage-cost record of the node's FIRST outgoing edge (that edge's record is evaluated at exactly (x_k^s, u_k^s), csrc/dompc_edge.h
      eval_models kind 1) - nodes of stages 0 .. N-1;
    * the terminal-cost record of the leaf's incoming edge (kind 2) - nodes of stage N.

    The kernels weight both records with the edge's omega (_mpc.py:1259-1261); an added term has no such weight in the reference, so it is
    divided by omega here.  lowering.lower_model generates one function per DISTINCT term (hash-consed: the same expression at several
    nodes is one function) and an edge -> function table."""

    def __init__(self, mpc):
        self.mpc = mpc
        ps = mpc.structure
        self.cx = [sym.symbol("xtra_xs%d" % i) for i in range(ps.nx)]
        self.cu = [sym.symbol("xtra_us%d" % i) for i in range(ps.nu)]
        self.cw = [sym.symbol("xtra_w%d" % i) for i in range(ps.M * ps.nx)]       # collocation states of an interval, slot-major
        self.cP = {}
        self.groups = {}            # (kind, edge) -> Node (filled by tables())
        self._parts = {}            # (kind, edge) -> {addend idx: [factor, addend in canonical symbols]}
        self.const_terms = []
        self._dummy = set(int(g) for g in ps.tables["dummy_idx"])

    def add(self, atom, scale):
        """returns None (accepted) or the reason for a refusal"""
        mpc, ps = self.mpc, self.mpc.structure
        T = ps.tables
        ox, op = mpc.opt_x, mpc.opt_p
        free = sym.free_symbols([atom])
        sx_ = [(ox.index_of[id(n)], n) for n in free if id(n) in ox.index_of]
        sp_ = [(op.index_of[id(n)], n) for n in free if id(n) in op.index_of]
        if not sx_:
            self.const_terms.append(sym.SX([sym.mul(sym.const(scale), atom)], (1, 1)))
            return None
        ix = sorted(g for g, _ in sx_)
        c = classify(mpc, sym.SX([atom], (1, 1)))
        if any(g in self._dummy for g in ix):
            return ("it depends on unused entries of the reference's opt_x (%s): no node of the scenario tree owns them"
                    % describe_variables(mpc, [g for g in ix if g in self._dummy]))
        if c["interval_unknowns"]:
            # collocation states `_x[k, s, c]`, c < M, of ONE interval (the reference's docstring example puts its terminal cost on all stored
            # points of the last interval): they are unknowns of the edge INTO node (k, s) - the term joins that edge's block on the dense
            # edge path (kind "ew": gradient / Hessian shares over the edge's own unknowns, csrc/dompc_dae.h)
            M, nx = ps.M, ps.nx
            slots = set()
            for g in ix:
                if g >= ps.off_z:
                    slots = None
                    break
                kk, r = divmod(g, ps.S * (1 + M) * nx)
                ss, r = divmod(r, (1 + M) * nx)
                if r // nx == M:
                    slots = None
                    break
                slots.add((kk, ss))
            if slots is None or len(slots) != 1 or ps.nz or ps.open_loop_stack or ps.eps_global or ps.M * ps.nx > 64:
                return ("the addend over (%s) couples collocation states of an interval with other variables, or touches algebraic / slack "
                        "unknowns: only terms in the collocation states `_x[k, s, c]` of ONE interval, or in the node state `_x[k, s, -1]` / "
                        "input `_u[k, s]` of ONE node, are lowered" % describe_variables(mpc, ix))
            (k, s), = slots
            n = int(T["level_node_start"][k]) + s
            e = int(T["node_in_edge"][n])
            w0 = int(T["edge_w_off"][e])
            mapping = {nd.idx: self.cw[g - w0] for g, nd in sx_}
            self._add(("ew", e), atom, scale, mapping, sp_)
            return None
        if len(c["nodes"]) > 1:
            return ("one addend couples %d nodes of the scenario tree (%s): the Riccati recursion eliminates one node at a time"
                    % (len(c["nodes"]), describe_variables(mpc, ix)))
        if ps.open_loop_stack or ps.eps_global:
            return "added cost terms are not lowered for open_loop with several scenarios / nl_cons_single_slack"
        k, s = c["nodes"][0]
        n = int(T["level_node_start"][k]) + s
        if k == ps.N:
            kind, e = "mt", int(T["node_in_edge"][n])
        else:
            kind, e = "lt", int(T["node_child_start"][n])
        x0, u0 = int(T["node_x_off"][n]), int(T["node_u_off"][n])
        mapping = {}
        for g, nd in sx_:
            if x0 <= g < x0 + ps.nx:
                mapping[nd.idx] = self.cx[g - x0]
            else:
                assert u0 >= 0 and u0 <= g < u0 + ps.nu, (g, x0, u0)
                mapping[nd.idx] = self.cu[g - u0]
        self._add((kind, e), atom, scale, mapping, sp_)
        return None

    def _add(self, key, atom, scale, mapping, sp_):
        for j, nd in sp_:
            if j not in self.cP:
                self.cP[j] = sym.symbol("xtra_P%d" % j)
            mapping[nd.idx] = self.cP[j]
        # addends of a node / an interval are collected with their factors and summed in a canonical order (tables): the same terms at two
        # nodes become the same expression node - one device function - in whatever order the user added them
        can = sym.substitute_nodes([atom], mapping)[0]
        slot = self._parts.setdefault(key, {}).setdefault(can.idx, [0.0, can])
        slot[0] += scale / float(self.mpc.structure.tables["edge_omega"][key[1]])
        self.groups[key] = None

    def _sum(self, key):
        total = None
        for coef, can in sorted(self._parts[key].values(), key=lambda cv: (cv[1].skey, cv[1].idx)):
            term = sym.mul(sym.const(coef), can)
            total = term if total is None else sym.add(total, term)
        return total

    def tables(self):
        """(lt_exprs, mt_exprs, lt_id[E], mt_id[E]): distinct expressions per kind and the 1-based function index of every edge (0: none)"""
        E = self.mpc.structure.n_edges
        out = {}
        for key in self._parts:
            self.groups[key] = self._sum(key)
        for kind in ("lt", "mt", "ew"):
            exprs, ids, seen = [], np.zeros(E, np.int32), {}
            for (kd, e), nd in sorted(self.groups.items(), key=lambda kv: kv[0][1]):
                if kd != kind:
                    continue
                if nd.idx not in seen:
                    exprs.append(nd)
                    seen[nd.idx] = len(exprs)
                ids[e] = seen[nd.idx]
            out[kind] = (exprs, ids)
        return out


class ConstraintExtras:
    """Inequality rows appended to `nlp_cons` (optimizer.py:131-215) that stay inside ONE node of the tree:  lb <= h(x_k^s, u_k^s; opt_p) <= ub,
    k < N.  The kernels give every edge the same number of nl_cons rows, each with a slack variable s (d(x) - s = 0, lb <= s <= ub, IPOPT's
    treatment of inequality rows); an added row takes one of `n_slots` EXTRA row slots of the node's first outgoing edge - that edge's rows
    are evaluated at exactly (x_k^s, u_k^s).  On all other edges the slot is MASKED: the row function is identically zero and its slack has no
    bounds, which makes the row inert in every formula of the algorithm (residual 0, Jacobian 0, Sigma_s = 0, multiplier 0, no barrier term,
    no step-size limit); the one place where the NUMBER of rows enters - the scaling s_d of the dual infeasibility, (|y|_1 + |z|_1) / (m + n) -
    gets the masked rows subtracted (csrc/dompc_driver.h: DOMPC_XROW_MASKED).  The solver works on an internal row layout
    (structure with ne + n_slots rows per edge); solver.RowMappedSolver translates bounds in and g / lam_g out, so the user sees the
    reference's order: structured rows, then the appended rows in the order they were appended."""

    def __init__(self, mpc):
        self.mpc = mpc
        ps = mpc.structure
        self.cx = [sym.symbol("xrow_xs%d" % i) for i in range(ps.nx)]
        self.cu = [sym.symbol("xrow_us%d" % i) for i in range(ps.nu)]
        self.cP = {}
        self.rows = []              # (edge, slot, canonical expression, lb, ub) in the order they were appended
        self._per_edge = {}
        self._dummy = set(int(g) for g in ps.tables["dummy_idx"])

    @property
    def n_slots(self):
        return max(self._per_edge.values()) if self._per_edge else 0

    def add(self, node, lb, ub):
        mpc, ps = self.mpc, self.mpc.structure
        T = ps.tables
        c = classify(mpc, sym.SX([node], (1, 1)))
        ix = c["opt_x"]
        if not ix:
            return "the row does not depend on any optimisation variable"
        if any(g in self._dummy for g in ix):
            return ("it depends on unused entries of the reference's opt_x (%s): no node of the scenario tree owns them"
                    % describe_variables(mpc, [g for g in ix if g in self._dummy]))
        if len(c["nodes"]) > 1:
            return ("it couples %d nodes of the scenario tree (%s): the Riccati recursion eliminates one node at a time"
                    % (len(c["nodes"]), describe_variables(mpc, ix)))
        if c["interval_unknowns"]:
            return ("it depends on collocation / algebraic / slack unknowns of an interval (%s), which are eliminated inside the "
                    "interval's own constraint block" % describe_variables(mpc, ix))
        k, s = c["nodes"][0]
        if k == ps.N:
            return ("a row in the state of a leaf (%s) has no outgoing edge whose row block could carry it - use the terminal bounds "
                    "(mpc.terminal_bounds) or a row on stage N - 1" % describe_variables(mpc, ix))
        if not (lb < ub):
            return ("an EQUALITY row (lb = ub = %g) at %s: the kernels carry added rows as inequality rows with a slack variable"
                    % (lb, describe_variables(mpc, ix)))
        if not (np.isfinite(lb) or np.isfinite(ub)):
            return "a row without any finite bound"
        if ps.open_loop_stack or ps.eps_global or ps.nz or getattr(mpc, "_nl_colloc", False) or getattr(mpc, "_estimator_opts", None):
            return ("added rows are not lowered for open_loop with several scenarios, nl_cons_single_slack, models with algebraic states, "
                    "nl_cons_check_colloc_points and estimators")
        n = int(T["level_node_start"][k]) + s
        e = int(T["node_child_start"][n])
        x0, u0 = int(T["node_x_off"][n]), int(T["node_u_off"][n])
        ox, op = mpc.opt_x, mpc.opt_p
        mapping = {}
        for nd in sym.free_symbols([node]):
            if id(nd) in ox.index_of:
                g = ox.index_of[id(nd)]
                mapping[nd.idx] = self.cx[g - x0] if x0 <= g < x0 + ps.nx else self.cu[g - u0]
            else:
                j = op.index_of[id(nd)]
                if j not in self.cP:
                    self.cP[j] = sym.symbol("xrow_P%d" % j)
                mapping[nd.idx] = self.cP[j]
        slot = self._per_edge.get(e, 0)
        self._per_edge[e] = slot + 1
        self.rows.append((e, slot, sym.substitute_nodes([node], mapping)[0], float(lb), float(ub)))
        return None

    def row_map(self, ps_ref, ps_int):
        """index of every row of the reference's g (structured rows, then the appended ones) inside the internal layout"""
        rpe = ps_ref.rows_per_edge
        m = np.empty(ps_ref.n_g + len(self.rows), np.int64)
        m[:ps_ref.nx] = np.arange(ps_ref.nx)
        r0_ref, r0_int = ps_ref.tables["edge_row0"], ps_int.tables["edge_row0"]
        for e in range(ps_ref.n_edges):
            m[r0_ref[e]:r0_ref[e] + rpe] = r0_int[e] + np.arange(rpe)
        for j, (e, slot, _, _, _) in enumerate(self.rows):
            m[ps_ref.n_g + j] = r0_int[e] + rpe + slot
        return m


def check_additions(mpc) -> None:
    """create_nlp(): everything the user added after prepare_nlp() is classified; the structured backend refuses what it cannot lower."""
    obj, cons, lbs, ubs = mpc._nlp_obj, mpc._nlp_cons, mpc._nlp_cons_lb, mpc._nlp_cons_ub
    if not isinstance(obj, NlpObjective):
        raise NotImplementedError("structured HIP backend: nlp_obj was replaced by %r; the structured objective can only be extended "
                                  "(nlp_obj += expression)" % (type(obj).__name__,))
    if not (isinstance(cons, list) and cons and isinstance(cons[0], StructuredBlock)):
        raise NotImplementedError("structured HIP backend: the structured constraint block (first entry of nlp_cons) was removed or "
                                  "replaced; append to nlp_cons / nlp_cons_lb / nlp_cons_ub instead")
    if not (isinstance(lbs, list) and isinstance(ubs, list)):
        # the setters take anything, as the reference's do (optimizer.py:172-215); an array in place of the list of blocks cannot be matched
        # to nlp_cons block by block, so it is only accepted when nothing was appended
        if len(cons) != 1:
            raise ValueError("nlp_cons_lb / nlp_cons_ub must stay lists with one entry per nlp_cons block (%d blocks); got %s / %s"
                             % (len(cons), type(lbs).__name__, type(ubs).__name__))
        lbs = mpc._nlp_cons_lb = lbs if isinstance(lbs, list) else [lbs]
        ubs = mpc._nlp_cons_ub = ubs if isinstance(ubs, list) else [ubs]
    if not (len(cons) == len(lbs) == len(ubs)):
        raise ValueError("nlp_cons, nlp_cons_lb and nlp_cons_ub must have one entry per constraint block: %d / %d / %d"
                         % (len(cons), len(lbs), len(ubs)))
    base_lb, base_ub = np.asarray(lbs[0], float).reshape(-1), np.asarray(ubs[0], float).reshape(-1)
    if base_lb.size != mpc.structure.n_g or base_ub.size != mpc.structure.n_g:
        raise ValueError("the bounds of the structured constraint block must keep their %d entries" % mpc.structure.n_g)
    problems = []
    extras = ObjectiveExtras(mpc)
    for j, ex in enumerate(obj.terms):
        # the objective is a SUM: every addend of an added expression is classified on its own (the reference's example
        # `sum1(vertcat(*opt_x['_x', -1, 0]) ** 2)`, optimizer.py:91-97, is one expression over several vectors)
        exs = sym._sx(ex)
        if exs.numel() != 1:
            raise ValueError("nlp_obj term %d is not a scalar expression (shape %s)" % (j, exs.shape))
        c = classify(mpc, exs)
        if c["foreign"]:
            raise ValueError("nlp_obj term %d uses symbols that belong neither to mpc.opt_x nor to mpc.opt_p: %s"
                             % (j, ", ".join(repr(n) for n in c["foreign"][:4])))
        for scale, atom in split_additive(exs.nodes()[0]):
            why = extras.add(atom, scale)
            if why:
                problems.append("nlp_obj term %d: %s" % (j, why))
    rows = ConstraintExtras(mpc)
    for j, ex in enumerate(cons[1:]):
        what = "nlp_cons block"
        c = classify(mpc, ex)
        if c["foreign"]:
            raise ValueError("%s %d uses symbols that belong neither to mpc.opt_x nor to mpc.opt_p: %s"
                             % (what, j, ", ".join(repr(n) for n in c["foreign"][:4])))
        exs = sym._sx(ex)
        lb_j, ub_j = np.asarray(lbs[1 + j], float).reshape(-1), np.asarray(ubs[1 + j], float).reshape(-1)
        if lb_j.size != exs.numel() or ub_j.size != exs.numel():
            raise ValueError("%s %d has %d rows, its bounds %d / %d entries" % (what, j, exs.numel(), lb_j.size, ub_j.size))
        # every ROW is classified on its own: a block may hold rows of several nodes
        for q, nd in enumerate(exs.nodes()):
            why = rows.add(nd, lb_j[q], ub_j[q])
            if why:
                problems.append("%s %d, row %d: %s" % (what, j, q, why))
    if problems:
        raise NotImplementedError("structured HIP backend: the NLP was modified after prepare_nlp() in a way that is no longer "
                                  "stage-structured -\n  " + "\n  ".join(problems) +
                                  "\n(the reference hands such an NLP to CasADi/IPOPT as one sparse problem, "
                                  "/root/reference/do_mpc/optimizer.py:1050-1094; this backend has no general sparse fallback)")
    # accepted: (a) constant objective terms (functions of opt_p only) - kept for inspection; the objective value this backend reports is
    # the structured objective WITHOUT them (u0 and every other solution quantity are unaffected); (b) node-local cost terms - lowered
    # into per-node device functions by lowering.lower_model (`extras`)
    mpc._nlp_obj_const_terms = extras.const_terms
    mpc._nlp_extras = extras if extras.groups else None
    mpc._nlp_rows = rows if rows.rows else None
