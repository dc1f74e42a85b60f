"""Tree sharding of ONE problem over several ranks: the sub-trees below the cut of the
scenario tree go to the ranks, the stages above are replicated, and the ranks exchange cut-edge
contributions of the Riccati recursion plus the scalar reductions of the IPM as element-wise SUMs.

CPU coverage of the multi-rank path: the kernel text runs in the host emulation (tests/hostemu.py),
the collective is torch.distributed all_reduce on a gloo group with world_size 2 and 3.  The sharded
solves must reproduce the single-rank solve of the same problem."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import hostemu
from do_mpc_amd.examples import CASES
from do_mpc_amd.structure import build_structure, shard_tables

PAIRED = {"n_robust": 2, "uncertainty": "paired"}


def test_masks_partition_variables_rows_and_edges():
    ps = build_structure(nx=10, nu=3, nz=0, np_=2, ntvp=0, ne=1, ns=1, deg=2, ni=1, N=20, n_comb=3, n_robust=3, discrete=False)
    for world in (2, 4, 8):
        own_x = np.zeros(ps.n_opt_x, int)
        own_g = np.zeros(ps.n_g, int)
        own_e = np.zeros(ps.n_edges, int)
        for r in range(world):
            t = shard_tables(ps, r, world)
            own_x += t["x_mask"] == 1
            own_g += t["g_mask"] == 1
            own_e += t["edge_mask"] == 1
            rep = (t["x_mask"] == 2, t["g_mask"] == 2, t["edge_mask"] == 2)
            assert t["n_cut"] == ps.scenario_tree["n_scenarios"][t["cut_level"] - 1]
            assert ps.scenario_tree["n_scenarios"][t["cut_level"]] >= world
            # a cut parent is replicated and all its children are sub-tree roots
            for n in np.where(t["node_cut"] >= 0)[0]:
                assert t["node_mask"][n] == 2
                cs, cc = ps.tables["node_child_start"][n], ps.tables["node_child_count"][n]
                assert all(t["node_mask"][ps.tables["edge_child"][cs + j]] in (0, 1) for j in range(cc))
        assert np.all(own_x + rep[0] == 1) and np.all(own_g + rep[1] == 1) and np.all(own_e + rep[2] == 1)


def _solve(name, kw, shard=None):
    ex = CASES[name]
    with hostemu.patched():
        mpc = ex.build_mpc(ex.build_model(), **kw)
        mpc.x0 = ex.X0
        mpc.set_initial_guess()
        info = mpc.shard_tree(**shard) if shard else None
        u0 = mpc.make_step(ex.X0).ravel().copy()
        return (u0, mpc.opt_x_num.master.copy(), np.array(mpc.lam_g_num).copy(), dict(mpc.solver_stats), info,
                mpc.structure.tables["dummy_idx"])


@pytest.mark.parametrize("name,kw,cut", [("CSTR", {}, 1), ("industrial_poly", PAIRED, 1), ("industrial_poly", PAIRED, 2)])
def test_forced_cut_on_one_rank_is_the_unsharded_solve(name, kw, cut):
    """world = 1 with a cut: every exchange is the identity, the cut-parent code path must give the same iterates."""
    u_ref, x_ref, lg_ref, st_ref, _, dummy = _solve(name, kw)
    calls = []
    u, x, lg, st, info, _ = _solve(name, kw, dict(rank=0, world=1, cut_level=cut, allreduce=lambda v: calls.append(v.numel())))
    assert info["cut_level"] == cut and len(calls) > st["iter_count"]
    assert st["success"] and abs(st["iter_count"] - st_ref["iter_count"]) <= 1
    keep = np.ones(x.size, bool)
    keep[dummy] = False                                    # variables in no row / cost term: not determined
    # several cut parents: the sums are formed in another order -> the iterates differ at rounding level and the
    # run may stop one iteration earlier or later (both points satisfy the 1e-8 tolerances)
    assert np.allclose(u, u_ref, rtol=1e-7, atol=0)
    assert np.allclose(x[keep], x_ref[keep], rtol=1e-6, atol=1e-8)
    assert np.max(np.abs(lg - lg_ref) / np.maximum(1.0, np.abs(lg_ref))) < 1e-5


def test_forced_cut_with_cost_terms_added_to_nlp_obj():
    """the sharding-aware code object of an NLP that was extended on the low-level route (cost terms at leaves, at an inner node, at the
    root: per-node device functions joined to the edges' cost records) - the records of owned and replicated edges alike"""
    import route_cases as rc
    ex = CASES["industrial_poly"]
    sol = []
    for shard in (None, dict(rank=0, world=1, cut_level=2, allreduce=lambda v: None)):
        with hostemu.patched():
            mpc = rc.stopped_before_setup(lambda n: ex.build_mpc(ex.build_model(), **PAIRED), "industrial_poly")
            mpc.prepare_nlp()
            rc.ADDED_COST["tree"](mpc)
            mpc.create_nlp()
            mpc.x0 = ex.X0
            mpc.set_initial_guess()
            if shard:
                mpc.shard_tree(**shard)
            mpc.make_step(ex.X0)
        assert mpc.solver_stats["success"] and "#define DOMPC_XTRA 1" in mpc.generated_header
        sol.append((mpc.solver_stats["iter_count"], mpc.opt_x_num.master.copy(), mpc.structure.tables["dummy_idx"]))
    keep = np.ones(sol[0][1].size, bool)
    keep[sol[0][2]] = False
    assert abs(sol[0][0] - sol[1][0]) <= 1
    assert np.allclose(sol[0][1][keep], sol[1][1][keep], rtol=1e-6, atol=1e-8)


def _worker(rank, world, port, name, kw, cut, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        u, x, lg, st, info, dummy = _solve(name, kw, dict(rank=rank, world=world, cut_level=cut))
        q.put((rank, u, x, lg, st["iter_count"], st["success"], int((info["edge_mask"] == 1).sum())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,kw,world,cut", [("CSTR", {}, 2, None), ("industrial_poly", PAIRED, 2, None),
                                               ("industrial_poly", PAIRED, 3, 2)])
def test_gloo_ranks_reproduce_the_single_rank_solve(name, kw, world, cut):
    u_ref, x_ref, lg_ref, st_ref, _, dummy = _solve(name, kw)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, kw, cut, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    keep = np.ones(x_ref.size, bool)
    keep[dummy] = False
    assert sum(r[6] for r in res) > 0 and all(r[6] > 0 for r in res)          # every rank owns edges
    for rank, u, x, lg, iters, ok, _ in res:
        assert ok and abs(iters - st_ref["iter_count"]) <= 2
        assert np.allclose(u, u_ref, rtol=1e-7, atol=0), (rank, u, u_ref)
        assert np.allclose(x[keep], x_ref[keep], rtol=1e-6, atol=1e-8)
        # (the ranks may stop one iteration earlier or later than the single-rank run - both points satisfy the 1e-8
        #  tolerances; the multipliers of the two acceptable points differ by a few 1e-5)
        assert np.max(np.abs(lg - lg_ref) / np.maximum(1.0, np.abs(lg_ref))) < 2e-4
    # all ranks hold the same combined solution
    for r in res[1:]:
        assert np.array_equal(r[1], res[0][1]) and np.array_equal(r[2][keep], res[0][2][keep])


def test_shard_description_is_validated():
    """dompc_set_sharding refuses inconsistent descriptions instead of running with them (C ABI error path)."""
    import ctypes as C
    from do_mpc_amd.solver import ShardDesc, _ALLREDUCE_FN
    ex = CASES["CSTR"]
    with hostemu.patched():
        mpc = ex.build_mpc(ex.build_model())
    S = mpc.S
    t = shard_tables(mpc.structure, 0, 2)
    n = int(S._lib.dompc_exchange_doubles(S._h, 2, t["n_cut"]))
    assert n == 2 * 12 + t["n_cut"] * (S._lib.dompc_exchange_doubles(S._h, 1, 1) - 12 - 2) + 2 * 2   # [W x RED_MAX | per cut parent | 2W flags]
    buf = np.zeros(n)
    keep = {k: np.ascontiguousarray(t[k]) for k in ("x_mask", "g_mask", "edge_mask", "node_mask", "node_cut")}
    cb = _ALLREDUCE_FN(lambda ctx, b, c: None)

    def desc(**over):
        kw = dict(rank=0, world=2, cut_level=t["cut_level"], n_cut=t["n_cut"], xbuf=buf.ctypes.data, xbuf_doubles=n, allreduce=cb,
                  ctx=None, **{k: v.ctypes.data for k, v in keep.items()})
        kw.update(over)
        return ShardDesc(**kw)

    assert S._lib.dompc_set_sharding(S._h, C.byref(desc())) == 0
    assert S._lib.dompc_set_sharding(S._h, None) == 0                              # off again
    for bad in (dict(rank=2), dict(world=0), dict(cut_level=0), dict(n_cut=0), dict(xbuf_doubles=n - 1), dict(x_mask=None),
                dict(xbuf=None)):
        assert S._lib.dompc_set_sharding(S._h, C.byref(desc(**bad))) != 0, bad
        assert S._lib.dompc_last_error(S._h)
    # host emulation has no RCCL: the native collective is refused with a message
    raw = (C.c_uint8 * 128)()
    assert S._lib.dompc_rccl_unique_id(S._h, b"", raw) != 0 and b"RCCL" in S._lib.dompc_last_error(S._h)


@pytest.mark.parametrize("seed", range(6))
def test_random_trees_and_worlds_partition(seed):
    """ownership is a partition for arbitrary (n_comb, n_robust, N, world, cut_level): every variable / row / edge is
    owned by exactly one rank or replicated on all, sub-trees are contiguous, cut parents sit right above the cut"""
    rng = np.random.default_rng(seed)
    n_comb = int(rng.integers(2, 5))
    n_robust = int(rng.integers(1, 4))
    N = int(n_robust + rng.integers(0, 4))
    ps = build_structure(nx=int(rng.integers(1, 4)), nu=int(rng.integers(1, 3)), nz=0, np_=1, ntvp=0, ne=int(rng.integers(0, 2)),
                         ns=0, deg=2, ni=int(rng.integers(1, 3)), N=N, n_comb=n_comb, n_robust=n_robust, discrete=False)
    n_scen = ps.scenario_tree["n_scenarios"]
    cut = int(rng.integers(1, n_robust + 1))
    world = int(rng.integers(1, n_scen[cut] + 1))
    own = [np.zeros(n, int) for n in (ps.n_opt_x, ps.n_g, ps.n_edges, ps.n_nodes)]
    for r in range(world):
        t = shard_tables(ps, r, world, cut_level=cut)
        for acc, key in zip(own, ("x_mask", "g_mask", "edge_mask", "node_mask")):
            acc += t[key] == 1
        reps = [t[key] == 2 for key in ("x_mask", "g_mask", "edge_mask", "node_mask")]
        mine = np.where(t["node_mask"][ps.tables["level_node_start"][cut]:ps.tables["level_node_start"][cut + 1]] == 1)[0]
        assert mine.size == 0 or np.array_equal(mine, np.arange(mine[0], mine[-1] + 1))      # contiguous block of sub-tree roots
        lv = ps.tables["node_level"]
        assert np.all((t["node_cut"] >= 0) == (lv == cut - 1)) and np.all(t["node_mask"][lv < cut] == 2)
        # an owned node's whole sub-tree is owned: parent of an owned node below the cut is owned too
        par = ps.tables["node_parent"]
        below = np.where(lv > cut)[0]
        assert np.array_equal(t["node_mask"][below], t["node_mask"][par[below]])
    for acc, rep in zip(own, reps):
        assert np.all(acc + rep == 1)


def _worker27(rank, world, port, q):
    import parity_common as pc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ex = CASES["industrial_poly"]
        x0 = pc.golden("industrial_poly")["mpc._x"][0]
        with hostemu.patched():
            mpc = ex.build_mpc(ex.build_model(), **pc.TREE27)
            mpc.x0 = x0
            mpc.set_initial_guess()
            mpc.shard_tree(rank=rank, world=world, cut_level=2)
            u0 = mpc.make_step(x0).ravel().copy()
        q.put((rank, u0, mpc.opt_x_num.master.copy(), np.array(mpc.lam_g_num).copy(), dict(mpc.solver_stats)))
    finally:
        dist.destroy_process_group()


def test_mid_size_tree_on_three_gloo_ranks_against_the_oracle_solve():
    """27-leaf industrial_poly tree, cut level 2 (3 cut parents, 9 sub-trees) on three gloo ranks against the ORACLE's solve of
    the same problem (not only against the single-rank product): iteration count +-1, final iterate, multipliers."""
    import parity_common as pc
    nlp, x0, r = pc.oracle_tree27()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker27, args=(rk, 3, port, q)) for rk in range(3)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=900) for _ in range(3)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ex = CASES["industrial_poly"]
    with hostemu.patched():
        dummy = ex.build_mpc(ex.build_model(), **pc.TREE27).structure.tables["dummy_idx"]
    used = np.ones(nlp.n_opt_x, bool)
    used[dummy] = False
    for rank, u, x, lg, st in res:
        assert st["success"] and abs(st["iter_count"] - r["stats"]["iter_count"]) <= 1
        assert pc.relerr(u, nlp.u0_of(r["x"])) < 1e-6
        assert pc.relerr(x[used], r["x"][used]) < 1e-6
        assert np.max(np.abs(lg - r["lam_g"])) < 2e-4 * max(1.0, np.max(np.abs(r["lam_g"])))
