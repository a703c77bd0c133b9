"""Adaptive octree / quadtree partition (reference utils/adaptive_blocking.py:60-423, main.py:456-482)
without Gurobi.

The reference builds a full 2^dim-ary tree of depth `maxl`, prunes nodes whose data is (near-)constant
(variance <= var_thr and |mean| <= e_thr, together with all their descendants), gives every
remaining node the feature  max|FFT| / sum|FFT|  (cal_feature), and solves a binary program:

    maximise   sum_p  feature_p / (2^dim)^level_p * active_p
    s.t.       sum_p active_p <= Nb
               active_p = 0                                  for level_p < minl
               exactly one active node on every root->leaf chain that has no pruned node,
               at most one on chains that lost their tail to pruning (no row at all when fewer than
               two nodes of the chain survive).

That program is a tree knapsack.  `solve_tree` computes it exactly with a bottom-up DP
(best[node][k] = best objective of the node's subtree with k active nodes), so no solver is needed.

Parity pin (tests/golden/adaptive.npz, tests/test_adaptive_blocking.py): the reference's OctTree was run
with a recording stand-in for the gurobipy module; tree order, pruned-node set, every node's feature and
the objective coefficients are compared with the reference's own, and the DP's optimum VALUE with the
optimum of the reference's recorded program solved by an independent MILP solver (scipy / HiGHS).  An ILP
solver may return any optimal solution; the DP returns the one that prefers, on ties, fewer blocks and
then the node itself over its children.
The 2-D branch of the reference cannot execute as shipped (QuadTree.get_feature reads an undefined
self.Type, utils/adaptive_blocking.py:105; cal_feature has no branch for the 2-D gray patches it is handed,
:16-24), so the quadtree here follows the octree's semantics with 4^level weights and is unpinned.
"""
import math

import numpy as np

from .misc import cal_feature, chunk_name, rgb2gray

NEG = -1.0e300
GPU_FEATURE_VOXELS = 1 << 24     # volumes from 256^3 up take their octree statistics / FFT features on the GPU when there is one


class Node:
    __slots__ = ("level", "oz", "oy", "ox", "z", "y", "x", "d", "h", "w", "children", "pruned", "feature", "best", "choice")

    def __init__(self, level, oz, oy, ox):
        self.level, self.oz, self.oy, self.ox = level, oz, oy, ox
        self.children = []
        self.pruned = False
        self.feature = 0.0


def build_tree(shape, max_level, dim=3):
    """full tree; node (level, oz, oy, ox) covers a (d/2^l, h/2^l, w/2^l) box (dim 2: d = 1, oz = 0).  Child order
    matches Patch3d.get_children (z outer, y, x inner) / Patch2d.get_children (y outer, x inner)."""
    if dim == 3:
        d, h, w = shape[:3]
        assert d % (2 ** max_level) == 0, "image size error!"
    else:
        d, (h, w) = 1, shape[:2]
    assert h % (2 ** max_level) == 0 and w % (2 ** max_level) == 0, "image size error!"

    def make(level, oz, oy, ox):
        n = Node(level, oz, oy, ox)
        n.d = d // 2 ** level if dim == 3 else 1
        n.h, n.w = h // 2 ** level, w // 2 ** level
        n.z, n.y, n.x = n.d * oz, n.h * oy, n.w * ox
        if level < max_level:
            for i in range(2 if dim == 3 else 1):
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


def _block(data, n, dim):
    return data[n.z:n.z + n.d, n.y:n.y + n.h, n.x:n.x + n.w] if dim == 3 else data[n.y:n.y + n.h, n.x:n.x + n.w]


def _feature2d(gray):
    f = np.abs(np.fft.fft(np.fft.fft(gray, axis=0), axis=1))
    return int(f.max()) / int(f.sum())


def _gpu_stats(device):
    """(variance/mean, feature) of a block evaluated on the GPU in float64 — the same formulas as the numpy path (population
    variance; max|FFT| / sum|FFT| with both truncated to int).  At BASELINE config 4's size the reference's host version
    copies and transforms the 1024^3 root and every descendant (O(levels x volume) of complex128 numpy FFTs: minutes and tens
    of GiB); rocFFT does the four levels in seconds."""
    import torch

    def to_dev(blk):
        a = np.ascontiguousarray(blk)
        if not a.flags.writeable:        # a block of a read-only memory map: torch wants a buffer it may alias
            a = a.copy()
        if a.dtype == np.uint16:         # (moved as int16 bits, widened on the device: no float64 copy on the host)
            return (torch.from_numpy(a.view(np.int16)).to(device).to(torch.int32) & 0xFFFF).double()
        return torch.from_numpy(a).to(device).double()

    def var_mean(blk):
        t = to_dev(blk)
        m = t.mean()
        return float(((t - m) ** 2).mean().item()), float(m.item())

    def feature(blk):
        t = to_dev(blk)
        dims = (0, 1, 2) if t.ndim == 4 else (0, 1)
        f = torch.fft.fftn(t, dim=dims).abs()
        return int(f.max().item()) / int(f.sum().item())
    return var_mean, feature


def prune_and_score(root, data, var_thr=0.0, e_thr=0.0, feature_fn=None, dim=3, device=None):
    """OctTree.prune (:341-352) then get_feature (:289-292).  device: a torch device evaluates the block statistics and
    FFT features there (large volumes); None = numpy, the reference's arithmetic (what the goldens pin)."""
    if feature_fn is None:
        feature_fn = cal_feature if dim == 3 else _feature2d
    var_mean = None
    if device is not None:
        var_mean, feature_fn = _gpu_stats(device)

    def mark(n):
        n.pruned = True
        for c in n.children:
            mark(c)
    for n in iter_nodes(root):
        if n.pruned:
            continue
        blk = _block(data, n, dim)
        if var_mean is not None:
            v, m = var_mean(blk)
        else:
            m = blk.mean()
            v = ((blk - m) ** 2).mean()
        if v <= var_thr and abs(m) <= e_thr:
            mark(n)
    for n in iter_nodes(root):
        if not n.pruned:
            n.feature = float(feature_fn(_block(data, n, dim)))


def _maxplus(a, b, cap):
    """c[k] = max_{i+j=k} a[i]+b[j], with the smallest maximising i (k <= cap)"""
    n = cap + 1
    i = np.arange(n)[:, None]
    k = np.arange(n)[None, :]
    j = k - i
    s = np.where(j >= 0, a[:, None] + b[np.clip(j, 0, cap)], -np.inf)     # s[i, k]
    arg = np.argmax(s, axis=0)                                             # first maximum: smallest i
    c = s[arg, np.arange(n)]
    return np.where(c < NEG / 2, NEG, c), arg.astype(np.int64)


def solve_tree(root, Nb, min_level, dim=3):
    """exact DP; returns (active nodes in pre-order, objective value) or raises if infeasible"""
    cap = int(Nb)
    fan = float(2 ** dim)

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
            v = n.feature / (fan ** n.level)
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
    """main.py:456-482: (d,h,w,1|3) volumes through the octree, (h,w,3|1) images through the quadtree.
    Returns (chunk list, outline)."""
    _, _maxl, _minl, var_thr, e_thr, Nb = divide_type.split("_")
    var_thr, e_thr, Nb = int(var_thr), int(e_thr), int(Nb)
    if data.ndim == 4:
        dim = 3
        tree_data = data
        if data.shape[-1] == 3:                      # adaptive_cal_tree: per-slice RGB -> gray (:388-392)
            tree_data = rgb2gray(data, "rgb")[..., None]
        elif data.shape[-1] != 1:
            raise NotImplementedError("adaptive partition needs 1 or 3 channels")
    elif data.ndim == 3:
        dim = 2
        tree_data = rgb2gray(data, "rgb") if data.shape[-1] == 3 else data[..., 0]
    else:
        raise NotImplementedError("adaptive partition needs (d,h,w,c) or (h,w,c) data")
    Nb, minl, maxl = adaptive_levels(Nb, param_size, dim)
    root = build_tree(tree_data.shape, maxl, dim)
    device = None
    if tree_data.size >= GPU_FEATURE_VOXELS:
        import torch
        if torch.cuda.is_available():
            device = "cuda"
    prune_and_score(root, tree_data, var_thr, e_thr, dim=dim, device=device)
    active, _ = solve_tree(root, Nb, minl, dim)
    outline = data.copy()
    chunks = []
    for p in active:
        z, y, x, d, h, w = p.z, p.y, p.x, p.d, p.h, p.w
        if dim == 3:
            c = {"data": data[z:z + d, y:y + h, x:x + w], "d": [z, z + d - 1], "h": [y, y + h - 1], "w": [x, x + w - 1]}
            faces = ((z, slice(y, y + h), slice(x, x + w)), (z + d - 1, slice(y, y + h), slice(x, x + w)),
                     (slice(z, z + d), y, slice(x, x + w)), (slice(z, z + d), y + h - 1, slice(x, x + w)),
                     (slice(z, z + d), slice(y, y + h), x), (slice(z, z + d), slice(y, y + h), x + w - 1))
            for sl in faces:
                outline[sl] = 2000
        else:
            c = {"data": data[y:y + h, x:x + w], "h": [y, y + h - 1], "w": [x, x + w - 1]}
            col = np.array([0, 0, 255] if data.shape[-1] == 3 else [255], data.dtype)   # cv2.rectangle(..., (0,0,255), 2)
            for sl in ((slice(y, y + 2), slice(x, x + w)), (slice(y + h - 2, y + h), slice(x, x + w)),
                       (slice(y, y + h), slice(x, x + 2)), (slice(y, y + h), slice(x + w - 2, x + w))):
                outline[sl] = col
        c["total_size"], c["size"] = data.size, c["data"].size
        c["name"] = chunk_name(c)
        chunks.append(c)
    print("total numbers of the chunks: " + str(len(chunks)))
    return chunks, outline
