"""Adaptive octree partition without Gurobi: the tree-knapsack DP against exhaustive enumeration
of every feasible selection on small trees, plus cases whose optimum is forced.  (No golden
exists: the reference's solver needs a Gurobi licence.)"""
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
