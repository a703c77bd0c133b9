"""Adaptive octree partition (reference utils/adaptive_blocking.py:199-423, main.py:456-482)
without Gurobi.

The reference builds a full octree of depth `maxl`, prunes nodes whose data is (near-)constant
(variance <= var_thr and |mean| <= e_thr, together with all their descendants), gives every
remaining node the feature  max|FFT| / sum|FFT|  (cal_feature), and solves a binary program:

    maximise   sum_p  feature_p / 8^level_p * active_p
    s.t.       sum_p active_p <= Nb
               active_p = 0                                  for level_p < minl
               exactly one active node on every root->leaf chain that has no pruned node,
               at most one on chains that lost their tail to pruning.

That program is a tree knapsack.  `solve_tree` computes it exactly with a bottom-up DP
(best[node][k] = best objective of the node's subtree with k active nodes), so no solver is
needed.  Parity note: an ILP solver may return any optimal solution; the DP returns the one
that prefers, on ties, fewer blocks and then the node itself over its children.  No golden
vectors exist for this row (Gurobi is not available to run the reference's solver); the DP is
validated by brute force on small trees (tests/test_adaptive_blocking.py).
"""
import math

import numpy as np

from .misc import cal_feature, chunk_name

NEG = -1.0e300


class Node:
    __slots__ = ("level", "oz", "oy", "ox", "z", "y", "x", "d", "h", "w", "children", "pruned", "feature", "best", "choice")

    def __init__(self, level, oz, oy, ox):
        self.level, self.oz, self.oy, self.ox = level, oz, oy, ox
        self.children = []
        self.pruned = False
        self.feature = 0.0


def build_tree(shape, max_level):
    """full octree; node (level, oz, oy, ox) covers a (d/2^l, h/2^l, w/2^l) box.  Child order matches
    Patch3d.get_children (z outer, y, x inner)."""
    d, h, w = shape[:3]
    assert d % (2 ** max_level) == 0 and h % (2 ** max_level) == 0 and w % (2 ** max_level) == 0, "image size error!"

    def make(level, oz, oy, ox):
        n = Node(level, oz, oy, ox)
        n.d, n.h, n.w = d // 2 ** level, h // 2 ** level, w // 2 ** level
        n.z, n.y, n.x = n.d * oz, n.h * oy, n.w * ox
        if level < max_level:
            for i in range(2):
                for j in range(2):
                    for k in range(2):
                        n.children.append(make(level + 1, 2 * oz + i, 2 * oy + j, 2 * ox + k))
        return n
    return make(0, 0, 0, 0)


def iter_nodes(root):
    """pre-order, the order of OctTree.tree2list"""
    stack = [root]
    while stack:
        n = stack.pop()
        yield n
        stack.extend(reversed(n.children))


def prune_and_score(root, data, var_thr=0.0, e_thr=0.0, feature_fn=cal_feature):
    def mark(n):
        n.pruned = True
        for c in n.children:
            mark(c)
    for n in iter_nodes(root):
        if n.pruned:
            continue
        blk = data[n.z:n.z + n.d, n.y:n.y + n.h, n.x:n.x + n.w]
        m = blk.mean()
        if ((blk - m) ** 2).mean() <= var_thr and abs(m) <= e_thr:
            mark(n)
    for n in iter_nodes(root):
        if not n.pruned:
            n.feature = float(feature_fn(data[n.z:n.z + n.d, n.y:n.y + n.h, n.x:n.x + n.w]))


def _maxplus(a, b, cap):
    """c[k] = max_{i+j=k} a[i]+b[j], with the argmax i (k <= cap)"""
    n = cap + 1
    s = a[:, None] + b[None, :]                    # (i, j)
    c = np.full(n, NEG)
    arg = np.zeros(n, np.int64)
    for k in range(n):
        i = np.arange(0, k + 1)
        v = s[i, k - i]
        best = int(np.argmax(v))                   # first maximum: smallest i
        c[k], arg[k] = v[best], best
    return c, arg


def solve_tree(root, Nb, min_level):
    """exact DP; returns the list of active nodes (pre-order) or raises if infeasible"""
    cap = int(Nb)

    def solve(n):
        # best[k]: best value with exactly k actives in this subtree, NO ancestor active
        best = np.full(cap + 1, NEG)
        choice = {}
        if n.pruned:
            best[0] = 0.0
            n.best, n.choice = best, choice
            return
        for c in n.children:
            solve(c)
        # option (b): node inactive -> children cover themselves
        if n.children:
            acc = np.full(cap + 1, NEG)
            acc[0] = 0.0
            splits = []
            for c in n.children:
                acc, arg = _maxplus(acc, c.best, cap)
                splits.append(arg)
            comb, comb_splits = acc, splits
        else:
            comb, comb_splits = np.full(cap + 1, NEG), None      # an unpruned leaf must be active itself
        best = comb.copy()
        for k in range(cap + 1):
            if best[k] > NEG / 2:
                choice[k] = ("children", comb_splits)
        # option (a): node active (exactly one active on every chain through it)
        if n.level >= min_level and cap >= 1:
            v = n.feature / (8.0 ** n.level)
            if v >= best[1]:                       # tie -> the node itself
                best[1] = v
                choice[1] = ("self", None)
        n.best, n.choice = best, choice

    solve(root)
    feasible = [k for k in range(cap + 1) if root.best[k] > NEG / 2]
    if not feasible:
        raise ValueError("adaptive partition infeasible: Nb=%d is too small for min_level=%d" % (Nb, min_level))
    kbest = max(feasible, key=lambda k: (root.best[k], -k))
    active = []

    def collect(n, k):
        if n.pruned or k == 0 and not n.choice:
            return
        kind, splits = n.choice[k]
        if kind == "self":
            active.append(n)
            return
        # unwind the sequential max-plus merges
        ks = []
        rem = k
        for arg in reversed(splits):
            i = int(arg[rem])
            ks.append(rem - i)
            rem = i
        ks.reverse()
        for c, kc in zip(n.children, ks):
            collect(c, kc)

    collect(root, kbest)
    order = {id(n): i for i, n in enumerate(iter_nodes(root))}
    active.sort(key=lambda n: order[id(n)])
    return active, float(root.best[kbest])


def adaptive_levels(Nb, param_size, dimension=3):
    """utils/adaptive_blocking.py:395-416: Nb=-1 -> one block per 1361-parameter net; minl as uniform as
    possible, maxl two levels finer (the YAML's maxl/minl fields are ignored by the reference)."""
    if Nb == -1:
        Nb = int(param_size / (4 * 1361))
        if Nb <= 0:
            Nb = 1
    minl = math.floor(math.log(Nb, 2 ** dimension))
    return Nb, minl, minl + 2


def adaptive_chunk(data, param_size, divide_type):
    """main.py:456-482 for 3-D data (d,h,w,1): returns (chunk list, outline volume)"""
    _, _maxl, _minl, var_thr, e_thr, Nb = divide_type.split("_")
    var_thr, e_thr, Nb = int(var_thr), int(e_thr), int(Nb)
    if data.ndim != 4 or data.shape[-1] != 1:
        raise NotImplementedError("adaptive partition is implemented for single-channel 3-D data")
    Nb, minl, maxl = adaptive_levels(Nb, param_size, 3)
    root = build_tree(data.shape, maxl)
    prune_and_score(root, data, var_thr, e_thr)
    active, _ = solve_tree(root, Nb, minl)
    outline = data.copy()
    chunks = []
    for p in active:
        z, y, x, d, h, w = p.z, p.y, p.x, p.d, p.h, p.w
        c = {"data": data[z:z + d, y:y + h, x:x + w], "d": [z, z + d - 1], "h": [y, y + h - 1], "w": [x, x + w - 1]}
        c["total_size"], c["size"] = data.size, c["data"].size
        c["name"] = chunk_name(c)
        chunks.append(c)
        for sl in ((z, slice(y, y + h), slice(x, x + w)), (z + d - 1, slice(y, y + h), slice(x, x + w)),
                   (slice(z, z + d), y, slice(x, x + w)), (slice(z, z + d), y + h - 1, slice(x, x + w)),
                   (slice(z, z + d), slice(y, y + h), x), (slice(z, z + d), slice(y, y + h), x + w - 1)):
            outline[sl] = 2000
    print("total numbers of the chunks: " + str(len(chunks)))
    return chunks, outline
