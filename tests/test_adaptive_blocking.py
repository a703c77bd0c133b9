"""Adaptive octree partition without Gurobi: the tree-knapsack DP against exhaustive enumeration
of every feasible selection on small trees, cases whose optimum is forced, and the reference's own
octree / pruned set / features / binary program (tests/golden/adaptive.npz: utils/adaptive_blocking.py
run with a recording stand-in for gurobipy, its program solved by scipy's HiGHS)."""
import itertools

import numpy as np
import pytest

from brief_pytorch_amd import adaptive_blocking as ab
from brief_pytorch_amd.misc import merge_divided_data, parse_chunk_name
from brief_pytorch_amd.synthetic import make_volume


def enumerate_selections(n, min_level):
    """all (value, count) of feasible selections in n's subtree given no active ancestor"""
    if n.pruned:
        return [(0.0, 0)]
    out = []
    if n.level >= min_level:
        out.append((n.feature / 8.0 ** n.level, 1))
    if n.children:
        per_child = [enumerate_selections(c, min_level) for c in n.children]
        for combo in itertools.product(*per_child):
            out.append((sum(v for v, _ in combo), sum(k for _, k in combo)))
    return out


@pytest.mark.parametrize("seed,Nb,minl,prune_some", [(0, 8, 0, False), (1, 20, 1, False), (2, 12, 0, True), (3, 64, 1, True),
                                                      (4, 9, 1, True), (5, 1, 0, False), (6, 30, 2, False)])
def test_dp_matches_exhaustive(seed, Nb, minl, prune_some):
    rng = np.random.default_rng(seed)
    root = ab.build_tree((8, 8, 8), 2)
    nodes = list(ab.iter_nodes(root))
    for n in nodes:
        n.feature = float(rng.uniform(0.01, 1.0)) * (3.0 ** n.level)      # deeper blocks tend to pay off
    if prune_some:
        for n in rng.choice([m for m in nodes if m.level >= 1], size=5, replace=False):
            def mark(m):
                m.pruned = True
                for c in m.children:
                    mark(c)
            mark(n)
    feas = [(v, k) for v, k in enumerate_selections(root, minl) if k <= Nb]
    if not feas:
        with pytest.raises(ValueError):
            ab.solve_tree(root, Nb, minl)
        return
    best_v = max(v for v, _ in feas)
    active, val = ab.solve_tree(root, Nb, minl)
    assert abs(val - best_v) < 1e-12
    assert len(active) <= Nb and all(a.level >= minl and not a.pruned for a in active)
    assert abs(sum(a.feature / 8.0 ** a.level for a in active) - val) < 1e-12
    # exactly one active node on every unpruned root->leaf chain, at most one on pruned ones
    act = {id(a) for a in active}

    def check(n, above):
        here = above + (1 if id(n) in act else 0)
        if not n.children:
            assert here == 1 if not n.pruned else here <= 1
        for c in n.children:
            check(c, here)
    check(root, 0)


def test_forced_optimum_eight_octants():
    vol = make_volume((16, 16, 16), seed=11)
    Nb, minl, maxl = ab.adaptive_levels(8, 1e6)
    assert (Nb, minl, maxl) == (8, 1, 3)                 # Nb=8 -> exactly the eight octants are affordable
    chunks, outline = ab.adaptive_chunk(vol, 1e6, "adaptive_-1_-1_0_0_8")
    assert sorted(c["name"] for c in chunks) == sorted(
        "d_%d_%d-h_%d_%d-w_%d_%d" % (z, z + 7, y, y + 7, x, x + 7) for z in (0, 8) for y in (0, 8) for x in (0, 8))
    assert (outline == 2000).any()
    merged = merge_divided_data([{"data": c["data"], **parse_chunk_name(c["name"])} for c in chunks], list(vol.shape))
    assert np.array_equal(merged, vol)                   # a partition: every voxel exactly once


def test_pruned_region_is_left_out_and_levels():
    vol = make_volume((16, 16, 16), seed=12)
    vol[:8, :8, :8] = 0                                   # an all-zero octant is pruned (var<=0, |mean|<=0)
    chunks, _ = ab.adaptive_chunk(vol, 1e6, "adaptive_-1_-1_0_0_20")
    covered = np.zeros(vol.shape[:3], np.int32)
    for c in chunks:
        r = parse_chunk_name(c["name"])
        covered[r["d"][0]:r["d"][1] + 1, r["h"][0]:r["h"][1] + 1, r["w"][0]:r["w"][1] + 1] += 1
    assert covered[:8, :8, :8].max() == 0 and covered.max() == 1
    assert (covered[8:] == 1).all() and (covered[:, 8:] == 1).all() and (covered[:, :, 8:] == 1).all()
    assert len(chunks) <= 20
    assert ab.adaptive_levels(-1, 4 * 1361 * 70.5) == (70, 2, 4)
    assert ab.adaptive_levels(-1, 10.0) == (1, 0, 2)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_octree_program_and_optimum_match_the_reference(golden, tag):
    """tree order, pruned nodes, cal_feature of every node and the objective coefficients are the reference's;
    the DP's optimum equals the optimum of the reference's recorded binary program (any optimal solution has it)"""
    g = golden("adaptive")
    cfg = [int(v) for v in g[tag + "_cfg"]]
    shape, (seed, vthr, ethr, Nb, minl, maxl) = tuple(cfg[:3]), cfg[3:]
    vol = make_volume(shape, seed=seed)
    if tag + "_zbox" in g:
        zb = g[tag + "_zbox"]
        vol[zb[0][0]:zb[0][1], zb[1][0]:zb[1][1], zb[2][0]:zb[2][1]] = 0
    assert ab.adaptive_levels(Nb, 1e6) == (Nb, minl, maxl)
    root = ab.build_tree(vol.shape, maxl)
    ab.prune_and_score(root, vol, vthr, ethr)
    nodes = list(ab.iter_nodes(root))
    assert [[n.level, n.oz, n.oy, n.ox, int(n.pruned)] for n in nodes] == g[tag + "_nodes"].tolist()
    feats = np.array([0.0 if n.pruned else n.feature for n in nodes])
    assert np.allclose(feats, g[tag + "_features"], rtol=1e-12, atol=0)
    coef = np.array([0.0 if n.pruned else n.feature / 8.0 ** n.level for n in nodes])
    assert np.allclose(coef, g[tag + "_objcoef"], rtol=1e-12, atol=0)
    active, val = ab.solve_tree(root, Nb, minl)
    ref_val = float(g[tag + "_objval"][0])
    assert abs(val - ref_val) <= 1e-9 * abs(ref_val), (val, ref_val)
    # feasible in the reference's program (rows 1-4 of OctTree.solve_optim)
    assert len(active) <= Nb and all(a.level >= minl and not a.pruned for a in active)
    act = {id(a) for a in active}

    def check(n, above, chain_pruned):
        here = above + (1 if id(n) in act else 0)
        if not n.children:
            assert (here == 1) if not (chain_pruned or n.pruned) else here <= 1
        for c in n.children:
            check(c, here, chain_pruned or n.pruned)
    check(root, 0, False)
    ref_active = sorted(map(tuple, g[tag + "_active"].tolist()))
    mine = sorted((a.level, a.oz, a.oy, a.ox) for a in active)
    ref_sum = sum(coef[i] for i, n in enumerate(nodes) if (n.level, n.oz, n.oy, n.ox) in set(ref_active))
    assert abs(ref_sum - ref_val) <= 1e-9 * abs(ref_val)
    assert mine == ref_active            # no ties in these cases: the optimum is unique


def test_quadtree_partition_of_an_rgb_image():
    """2-D data (h,w,3): quadtree with 4^level weights; covers every non-constant pixel exactly once"""
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, size=(64, 96, 3)).astype(np.uint8)
    img[:32, :48] = 0
    chunks, outline = ab.adaptive_chunk(img, 1e6, "adaptive_-1_-1_0_0_12")
    assert ab.adaptive_levels(12, 1e6, 2) == (12, 1, 3)
    cover = np.zeros(img.shape[:2], np.int32)
    for c in chunks:
        r = parse_chunk_name(c["name"])
        assert "d" not in r
        cover[r["h"][0]:r["h"][1] + 1, r["w"][0]:r["w"][1] + 1] += 1
        assert np.array_equal(c["data"], img[r["h"][0]:r["h"][1] + 1, r["w"][0]:r["w"][1] + 1])
    assert cover[:32, :48].max() == 0 and cover.max() == 1 and (cover[32:] == 1).all() and (cover[:, 48:] == 1).all()
    assert 1 <= len(chunks) <= 12 and outline.shape == img.shape
