"""Integer structure of the multi-stage NLP: scenario tree, variable/constraint layout.

Mirrors (as tables instead of symbolic loops):
  Optimizer._setup_scenario_tree   /root/reference/do_mpc/optimizer.py:998-1048
  opt_x / opt_p ordering           /root/reference/do_mpc/controller/_mpc.py:1126-1134, 1160-1165
  constraint ordering              /root/reference/do_mpc/controller/_mpc.py:1189-1245
  collocation slot layout          /root/reference/do_mpc/optimizer.py:905-935
The tables are what the HIP kernels index with; opt_x itself stays in the reference's
canonical order ("stage-major": x[k][s][slot][state]) so vectors cross the C-ABI unchanged.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List

import numpy as np


def radau_points(deg: int) -> List[float]:
    """Right Radau points on (0,1] (casadi.collocation_points(deg,'radau'))."""
    if deg == 1:
        return [1.0]
    from numpy.polynomial import legendre as L
    c = np.zeros(deg + 1)
    c[deg], c[deg - 1] = 1.0, -1.0      # P_d - P_{d-1} vanishes at the right-Radau nodes (incl. +1)
    r = np.sort(np.real(L.legroots(c)))
    r[-1] = 1.0
    return list((r + 1.0) / 2.0)


def legendre_points(deg: int) -> List[float]:
    from numpy.polynomial import legendre as L
    c = np.zeros(deg + 1)
    c[deg] = 1.0
    return list((np.sort(np.real(L.legroots(c))) + 1.0) / 2.0)


def lagrange_collocation(deg: int, kind: str):
    """tau (deg+1), C[r,j] = L_r'(tau_j), D[r] = L_r(1) for the Lagrange basis on {0}+points
    (/root/reference/do_mpc/optimizer.py:844-888).  Closed-form barycentric evaluation."""
    if kind == "radau":
        pts = radau_points(deg)
    elif kind == "legendre":
        pts = legendre_points(deg)
    else:
        raise Exception("Unknown collocation scheme")
    tau = np.array([0.0] + pts)
    n = deg + 1
    C = np.zeros((n, n))
    D = np.zeros(n)
    for r in range(n):
        others = [m for m in range(n) if m != r]
        denom = np.prod([tau[r] - tau[m] for m in others])
        D[r] = np.prod([1.0 - tau[m] for m in others]) / denom
        for j in range(n):
            acc = 0.0
            for q in others:
                acc += np.prod([tau[j] - tau[m] for m in others if m != q])
            C[r, j] = acc / denom
    return tau, C, D


@dataclass
class ProblemStructure:
    nx: int
    nu: int
    nz: int
    np_: int
    ntvp: int
    ne: int          # rows of nl_cons per edge
    ns: int          # slack (eps) entries per (stage, scenario)
    deg: int
    ni: int
    M: int           # stored collocation slots per interval (0: discrete)
    N: int
    S: int
    n_comb: int
    n_robust: int
    n_eps: int
    discrete: bool
    tables: Dict[str, np.ndarray] = field(default_factory=dict)
    eps_global: bool = False     # nl_cons_single_slack: the slack variables are shared by all stages
    S_u: int = 0                 # scenario slots of `_u` (_mpc.py:1112-1117: 1 with open_loop, else S); 0 = S
    open_loop_stack: bool = False   # open_loop with several scenarios: solved as a chain over the stacked scenario states (open_loop.py)

    # -- sizes
    @property
    def n_x_block(self):
        return (self.N + 1) * self.S * (self.M + 1) * self.nx

    @property
    def off_z(self):
        return self.n_x_block

    @property
    def n_z_block(self):
        return self.N * self.S * max(self.M, 1) * self.nz

    @property
    def off_u(self):
        return self.off_z + self.n_z_block

    @property
    def SU(self):
        return self.S_u or self.S

    @property
    def off_eps(self):
        return self.off_u + self.N * self.SU * self.nu

    @property
    def n_opt_x(self):
        return self.off_eps + self.n_eps * self.S * self.ns

    @property
    def p_off_tvp(self):
        return self.nx

    @property
    def p_off_p(self):
        return self.nx + (self.N + 1) * self.ntvp

    @property
    def p_off_uprev(self):
        return self.p_off_p + self.n_comb * self.np_

    @property
    def n_opt_p(self):
        return self.p_off_uprev + self.nu

    @property
    def MZ(self):
        """z slots of an interval (_mpc.py:1130: max(n_total_coll_points, 1))"""
        return max(self.M, 1) if self.nz else 0

    @property
    def rows_block(self):
        """rows of an interval's own constraint block (optimizer.py:943-983: collocation, algebraic, continuity rows;
        discrete models: the algebraic rows) = number of its eliminated unknowns"""
        return self.M * self.nx + self.MZ * self.nz

    @property
    def rows_per_edge(self):
        return self.rows_block + self.nx + self.ne

    def iz(self, k, s, c):
        return self.off_z + ((k * self.S + s) * max(self.M, 1) + c) * self.nz

    @property
    def n_g(self):
        return self.nx + self.n_edges * self.rows_per_edge

    @property
    def n_edges(self):
        return len(self.tables["edge_parent"])

    @property
    def n_nodes(self):
        return len(self.tables["node_x_off"])

    def ix(self, k, s, c):
        return ((k * self.S + s) * (self.M + 1) + c) * self.nx

    def iu(self, k, s):
        return self.off_u + (k * self.SU + (s if self.SU == self.S else 0)) * self.nu       # (open_loop: one input for all scenarios of a stage)

    def ieps(self, e, s):
        return self.off_eps + (e * self.S + s) * self.ns


def build_structure(nx, nu, nz, np_, ntvp, ne, ns, deg, ni, N, n_comb, n_robust, discrete,
                    open_loop=False, single_slack=False) -> ProblemStructure:
    if n_robust > N:
        raise Exception("n_robust must not exceed n_horizon")
    S = n_comb ** n_robust
    M = 0 if discrete else (deg + 1) * ni
    n_eps = 1 if single_slack else N
    ps = ProblemStructure(nx=nx, nu=nu, nz=nz, np_=np_, ntvp=ntvp, ne=ne, ns=ns, deg=deg, ni=ni, M=M, N=N, S=S,
                          n_comb=n_comb, n_robust=n_robust, n_eps=n_eps, discrete=discrete)
    # open_loop with several scenarios (_mpc.py:1112-1117, 1205-1206): `_u` has ONE scenario slot, every scenario of a stage applies the
    # same input.  The layout below is the reference's; the problem is not tree-structured and is solved as a chain over the stacked
    # scenario states (do_mpc_amd/open_loop.py) - these tables then describe the reference's variables / rows for the mapping only.
    ps.S_u = 1 if open_loop else S
    ps.open_loop_stack = bool(open_loop and S > 1)
    # nl_cons_single_slack (_mpc.py:1120-1123, 1228): one `_eps` entry per scenario slot for ALL stages.  The slacks then are no
    # decision variables of a node; the kernels take them out of the tree-structured part and solve for them by a Schur
    # complement (csrc/dompc_driver.h: EPS_GLOBAL) - one extra linear solve per slack variable and iteration.
    ps.eps_global = bool(single_slack and ns > 0 and N > 1)
    if ps.eps_global:
        if S * ns > 32:
            raise NotImplementedError("structured HIP backend: nl_cons_single_slack with {} shared slack variables "
                                      "(scenarios x slack entries); the kernels handle at most 32".format(S * ns))
        if n_robust >= N:
            raise NotImplementedError("structured HIP backend: nl_cons_single_slack with n_robust = n_horizon "
                                      "(slack entries of scenario slots that no node reads)")

    n_branches = [n_comb if k < n_robust else 1 for k in range(N)]
    n_scen = [n_comb ** min(k, n_robust) for k in range(N + 1)]
    level_start = np.zeros(N + 2, dtype=np.int32)
    for k in range(N + 1):
        level_start[k + 1] = level_start[k] + n_scen[k]
    n_nodes = int(level_start[-1])

    node_level = np.zeros(n_nodes, np.int32)
    node_x_off = np.zeros(n_nodes, np.int32)
    node_u_off = -np.ones(n_nodes, np.int32)
    node_eps_off = -np.ones(n_nodes, np.int32)
    node_child_start = np.zeros(n_nodes, np.int32)
    node_child_count = np.zeros(n_nodes, np.int32)
    node_parent = -np.ones(n_nodes, np.int32)
    node_in_edge = -np.ones(n_nodes, np.int32)
    e_parent, e_child, e_pidx, e_woff, e_row0, e_level, e_omega = [], [], [], [], [], [], []
    parent_scenario = -np.ones((N + 1, S), dtype=int)
    child_scenario = -np.ones((N, S, n_branches[0] if N else 1), dtype=int)
    branch_offset = -np.ones((N, S), dtype=int)
    structure_scenario = np.zeros((N + 1, S), dtype=int)
    rpe = ps.rows_per_edge
    e_zoff = []
    for k in range(N + 1):
        for s in range(n_scen[k]):
            n = level_start[k] + s
            node_level[n] = k
            node_x_off[n] = ps.ix(k, s, M)
            if k < N:
                node_u_off[n] = ps.iu(k, s)
                if ns:
                    node_eps_off[n] = ps.ieps(min(k, n_eps - 1), s)
    for k in range(N):
        cnt = 0
        for s in range(n_scen[k]):
            n = level_start[k] + s
            node_child_start[n] = len(e_parent)
            node_child_count[n] = n_branches[k]
            boff = 0 if (n_robust == 0 or k < n_robust) else s % n_branches[0]
            branch_offset[k][s] = boff
            for b in range(n_branches[k]):
                c = cnt
                cnt += 1
                child_scenario[k][s][b] = c
                structure_scenario[k][c] = s
                structure_scenario[k + 1][c] = s
                parent_scenario[k + 1][c] = s
                cn = level_start[k + 1] + c
                node_parent[cn] = n
                node_in_edge[cn] = len(e_parent)
                e_row0.append(nx + len(e_parent) * rpe)
                e_parent.append(n)
                e_child.append(cn)
                e_pidx.append(b + boff)
                e_woff.append(ps.ix(k + 1, c, 0))
                e_zoff.append(ps.iz(k, c, 0) if nz else 0)     # (_mpc.py:1213: the dynamics use `_z[k, child, :]`)
                e_level.append(k)
                e_omega.append(1.0 / n_scen[k + 1])

    ps.tables = dict(
        level_node_start=level_start, node_level=node_level, node_x_off=node_x_off, node_u_off=node_u_off,
        node_eps_off=node_eps_off, node_child_start=node_child_start, node_child_count=node_child_count,
        node_parent=node_parent, node_in_edge=node_in_edge,
        edge_parent=np.array(e_parent, np.int32), edge_child=np.array(e_child, np.int32),
        edge_pidx=np.array(e_pidx, np.int32), edge_w_off=np.array(e_woff, np.int32),
        edge_row0=np.array(e_row0, np.int32), edge_level=np.array(e_level, np.int32),
        edge_omega=np.array(e_omega, np.float64),
        edge_z_off=np.array(e_zoff, np.int32),       # (host-side only: the kernels derive it, dompc_dae.h:edge_zoff)
    )
    # variables that appear in no constraint and no cost term
    used = np.zeros(ps.n_opt_x, bool)
    for n in range(n_nodes):
        used[node_x_off[n]:node_x_off[n] + nx] = True
        if node_u_off[n] >= 0:
            used[node_u_off[n]:node_u_off[n] + nu] = True
        if node_eps_off[n] >= 0:
            used[node_eps_off[n]:node_eps_off[n] + ns] = True
    for w in e_woff:
        used[w:w + M * nx] = True
    if nz:
        for z in e_zoff:
            used[z:z + ps.MZ * nz] = True
    ps.tables["dummy_idx"] = np.where(~used)[0].astype(np.int32)
    ps.scenario_tree = {
        "structure_scenario": structure_scenario, "n_branches": n_branches, "n_scenarios": n_scen,
        "parent_scenario": parent_scenario, "branch_offset": branch_offset, "child_scenario": child_scenario,
    }
    return ps


def shard_tables(ps: ProblemStructure, rank: int = 0, world: int = 1, cut_level: int = None) -> dict:
    """Ownership tables for sharding the scenario tree of ONE problem over `world` ranks.

    The tree is cut above level `cut_level` (default: the first level with at least `world` nodes, at most
    n_robust): the sub-trees rooted at the level-`cut_level` nodes go to the ranks in contiguous blocks (siblings
    stay together), everything above is replicated on every rank.  An edge and its collocation unknowns belong to
    its child node, so the only quantities that cross ranks are (a) the condensed contributions of the cut edges
    to their parents (`cut parents`, level cut_level-1) in the Riccati recursion and (b) scalar reductions.

    masks: 0 = another rank's, 1 = mine, 2 = replicated (identical on every rank; counted once, by rank 0).
    cut_level 0 means "no cut" (everything mine)."""
    T = ps.tables
    n_scen = ps.scenario_tree["n_scenarios"]
    N = ps.N
    n_nodes, n_edges = ps.n_nodes, ps.n_edges
    if cut_level is None:
        cut_level = 0
        if world > 1:
            cut_level = next((k for k in range(1, ps.n_robust + 1) if n_scen[k] >= world), ps.n_robust)
            if cut_level == 0:
                raise ValueError("a tree without branching (n_robust = 0) cannot be sharded")
    if not (0 <= cut_level <= max(ps.n_robust, 0)):
        raise ValueError(f"cut_level {cut_level} outside 0..n_robust={ps.n_robust}")
    if world > 1 and cut_level == 0:
        raise ValueError("world > 1 needs a cut level >= 1")
    c = cut_level
    node_mask = np.ones(n_nodes, np.int8)
    node_cut = -np.ones(n_nodes, np.int32)
    n_cut = 0
    if c > 0:
        n_roots = n_scen[c]
        for n in range(n_nodes):
            k = int(T["node_level"][n])
            s = n - int(T["level_node_start"][k])
            if k < c:
                node_mask[n] = 2
                if k == c - 1:
                    node_cut[n] = s
            else:
                anc = s // (n_scen[k] // n_roots)
                node_mask[n] = 1 if (anc * world) // n_roots == rank else 0
        n_cut = n_scen[c - 1]
    edge_mask = node_mask[T["edge_child"]].astype(np.int8) if n_edges else np.zeros(0, np.int8)
    x_mask = np.full(ps.n_opt_x, 2 if c > 0 else 1, np.int8)
    nx, nu, ns, M = ps.nx, ps.nu, ps.ns, ps.M
    for n in range(n_nodes):
        m = node_mask[n]
        x_mask[T["node_x_off"][n]:T["node_x_off"][n] + nx] = m
        if T["node_u_off"][n] >= 0:
            x_mask[T["node_u_off"][n]:T["node_u_off"][n] + nu] = m
        if T["node_eps_off"][n] >= 0:
            x_mask[T["node_eps_off"][n]:T["node_eps_off"][n] + ns] = m
    rpe = ps.rows_per_edge
    g_mask = np.full(ps.n_g, 2 if c > 0 else 1, np.int8)
    for e in range(n_edges):
        w = int(T["edge_w_off"][e])
        x_mask[w:w + M * nx] = edge_mask[e]
        if ps.nz:
            z = int(T["edge_z_off"][e])
            x_mask[z:z + ps.MZ * ps.nz] = edge_mask[e]
        r0 = int(T["edge_row0"][e])
        g_mask[r0:r0 + rpe] = edge_mask[e]
    return dict(cut_level=c, n_cut=int(n_cut), rank=int(rank), world=int(world), node_mask=node_mask, edge_mask=edge_mask,
                node_cut=node_cut, x_mask=x_mask, g_mask=g_mask)
