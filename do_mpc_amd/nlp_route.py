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
    for what, items in (("nlp_obj term", obj.terms), ("nlp_cons block", cons[1:])):
        for j, ex in enumerate(items):
            c = classify(mpc, ex)
            if c["foreign"]:
                raise ValueError("%s %d uses symbols that belong neither to mpc.opt_x nor to mpc.opt_p: %s"
                                 % (what, j, ", ".join(repr(n) for n in c["foreign"][:4])))
            if c["constant"] and what == "nlp_obj term":
                continue        # a term in opt_p only shifts the objective: no effect on the solution (the reported f excludes it)
            why = []
            if len(c["nodes"]) > 1:
                why.append("it couples %d nodes of the scenario tree (%s): the Riccati recursion eliminates one node at a time"
                           % (len(c["nodes"]), describe_variables(mpc, c["opt_x"])))
            if c["interval_unknowns"]:
                why.append("it depends on collocation / algebraic / slack unknowns of an interval (%s), which are eliminated "
                           "inside the interval's own constraint block" % describe_variables(mpc, c["opt_x"]))
            if not why:
                k, s = c["nodes"][0] if c["nodes"] else (None, None)
                why.append("a node-specific %s at %s would need its own lowered device function for that node; the structured "
                           "lowering generates ONE function per kind (stage cost, terminal cost, nl_cons) for all nodes - express it "
                           "through set_objective / set_nl_cons / bounds (a time-varying weight in `_tvp` selects a stage)"
                           % ("cost term" if what == "nlp_obj term" else "constraint", describe_variables(mpc, c["opt_x"]) or "opt_p"))
            problems.append("%s %d (%d row%s): %s" % (what, j, c["rows"], "" if c["rows"] == 1 else "s", "; ".join(why)))
    if problems:
        raise NotImplementedError("structured HIP backend: the NLP was modified after prepare_nlp() in a way that is no longer "
                                  "stage-structured -\n  " + "\n  ".join(problems) +
                                  "\n(the reference hands such an NLP to CasADi/IPOPT as one sparse problem, "
                                  "/root/reference/do_mpc/optimizer.py:1050-1094; this backend has no general sparse fallback)")
    # accepted: constant objective terms (functions of opt_p only).  They are kept for inspection; the objective value this backend reports
    # is the structured objective WITHOUT them (u0 and every other solution quantity are unaffected)
    mpc._nlp_obj_const_terms = [sym._sx(t) for t in obj.terms]
