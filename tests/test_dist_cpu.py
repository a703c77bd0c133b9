"""N>1 path on CPU: two gloo ranks shard the blocks of a volume, each scores only its own blocks,
the all-reduced [SSE, n] must give the same PSNR as one process scoring everything."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from brief_pytorch_amd.dist_utils import allreduce_sse, assign_blocks
from brief_pytorch_amd.metrics import psnr_from_sse
from brief_pytorch_amd.misc import divide_data
from brief_pytorch_amd.synthetic import make_volume


def test_assign_blocks_lpt():
    costs = [8, 7, 6, 5, 4, 3, 2, 1]
    own = assign_blocks(costs, 2)
    loads = [sum(c for c, o in zip(costs, own) if o == r) for r in range(2)]
    assert sorted(own.count(r) for r in range(2)) == [4, 4] and abs(loads[0] - loads[1]) <= 1
    assert assign_blocks([1.0] * 8, 8) == list(range(8))            # C4: eight equal octants -> one per GPU
    assert assign_blocks(costs, 1) == [0] * 8
    mixed = [64, 16, 16, 16, 16, 4, 4, 4, 4, 4, 4, 4, 4]              # C5-like mixed block sizes
    own = assign_blocks(mixed, 4)
    loads = [sum(c for c, o in zip(mixed, own) if o == r) for r in range(4)]
    assert max(loads) == 64 and min(loads) >= 32


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    vol = make_volume((12, 16, 20), seed=5)
    rng = np.random.default_rng(0)
    dec = np.clip(vol.astype(np.int64) + rng.integers(-50, 50, size=vol.shape), 0, 65535).astype(np.uint16)
    chunks, _ = divide_data(vol, "total_2_2_2")
    dchunks, _ = divide_data(dec, "total_2_2_2")
    own = assign_blocks([float(c["size"]) for c in chunks], world)
    sse, cnt = 0.0, 0.0
    for i, (c, dc) in enumerate(zip(chunks, dchunks)):
        if own[i] == rank:
            d = c["data"].astype(np.int64) - dc["data"].astype(np.int64)
            sse += float((d * d).sum())
            cnt += float(c["size"])
    tot, n = allreduce_sse([sse], cnt, "cpu")
    full = vol.astype(np.int64) - dec.astype(np.int64)
    q.put((rank, float(tot[0]), n, float((full * full).sum()), float(vol.size), cnt))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sse_allreduce():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, tot, n, full, size, own_cnt in res:
        assert tot == full and n == size and 0 < own_cnt < size
        assert abs(psnr_from_sse(tot, n, 65535) - psnr_from_sse(full, size, 65535)) == 0


def _worker_eval(rank, world, port, q):
    """the collective pattern of NFGR._evaluate_divide without a GPU: rank 0 owns the partition and broadcasts the block
    list; every rank scores ITS z-slab of the merged volume; one all-reduce of [SSE, SSIM-sum, slices, voxels]"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from brief_pytorch_amd.dist_utils import allreduce_sum, broadcast_object
    from brief_pytorch_amd.metrics import ssim2d
    from brief_pytorch_amd.misc import merge_divided_data, parse_chunk_name
    dist.init_process_group("gloo", rank=rank, world_size=world)
    vol = make_volume((13, 24, 28), seed=6)                     # 13 slices over 3 ranks: ragged slabs
    rng = np.random.default_rng(1)
    dec = np.clip(vol.astype(np.int64) + rng.integers(-80, 80, size=vol.shape), 0, 65535).astype(np.uint16)
    desc = None
    if rank == 0:
        chunks, _ = divide_data(vol, "total_2_2_1")
        desc = [{"name": c["name"], "size": c["size"]} for c in chunks[:-1]]      # one block dropped: its region decodes as zeros
    desc = broadcast_object(desc)
    nz = vol.shape[0]
    z0, z1 = nz * rank // world, nz * (rank + 1) // world
    slab = np.zeros((z1 - z0,) + vol.shape[1:], np.uint16)
    for c in desc:
        r = parse_chunk_name(c["name"])
        za, zb = max(r["d"][0], z0), min(r["d"][1] + 1, z1)
        if za < zb:
            slab[za - z0:zb - z0, r["h"][0]:r["h"][1] + 1, r["w"][0]:r["w"][1] + 1] = dec[za:zb, r["h"][0]:r["h"][1] + 1, r["w"][0]:r["w"][1] + 1]
    d = slab.astype(np.int64) - vol[z0:z1].astype(np.int64)
    ss = sum(ssim2d(vol[z, ..., 0], slab[z - z0, ..., 0], 65535) for z in range(z0, z1))
    tot = allreduce_sum([float((d * d).sum()), ss, float(z1 - z0), float(d.size)], "cpu")
    # the max-intensity projections of the slabs, assembled by an elementwise MAX over the ranks (NFGR._evaluate_divide)
    from brief_pytorch_amd.dist_utils import allreduce_max
    md = allreduce_max(slab.max(0) if z1 > z0 else np.zeros(vol.shape[1:], np.uint16), "cpu")
    mh = np.zeros((nz,) + vol.shape[2:], np.uint16)
    mh[z0:z1] = slab.max(1)
    mh = allreduce_max(mh, "cpu")
    # what one process would have computed on the merged volume
    parts = [{"data": dec[parse_chunk_name(c["name"])["d"][0]:parse_chunk_name(c["name"])["d"][1] + 1,
                          parse_chunk_name(c["name"])["h"][0]:parse_chunk_name(c["name"])["h"][1] + 1,
                          parse_chunk_name(c["name"])["w"][0]:parse_chunk_name(c["name"])["w"][1] + 1], **parse_chunk_name(c["name"])} for c in desc]
    merged = merge_divided_data(parts, list(vol.shape))
    full = merged.astype(np.int64) - vol.astype(np.int64)
    ss_full = sum(ssim2d(vol[z, ..., 0], merged[z, ..., 0], 65535) for z in range(nz))
    assert md.dtype == np.uint16 and np.array_equal(md, merged.max(0)) and np.array_equal(mh, merged.max(1))
    q.put((rank, tot.tolist(), float((full * full).sum()), ss_full, nz, float(vol.size), len(desc)))
    dist.barrier()
    dist.destroy_process_group()


def test_three_rank_z_sharded_evaluation():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_eval, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(3)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, tot, sse_full, ss_full, nz, size, nblocks in res:
        assert nblocks == 5                                        # every rank got rank 0's block list (3 x 2 x 1 blocks of the ragged split, minus the dropped one)
        assert tot[0] == sse_full and tot[2] == nz and tot[3] == size
        assert abs(tot[1] - ss_full) < 1e-9


def test_eight_rank_z_sharded_evaluation_and_lpt():
    """the rank count of the SCALE run (8) on the CPU: the same collective pattern with 13 slices over 8 ranks (slabs of one or two
    slices), and the LPT assignment of a DivideTask's blocks to 8 ranks (C4: eight equal octants -> one each; C5-like mixes)"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_eval, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(8)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(8))
    for rank, tot, sse_full, ss_full, nz, size, nblocks in res:
        assert nblocks == 5 and tot[0] == sse_full and tot[2] == nz and tot[3] == size and abs(tot[1] - ss_full) < 1e-9
    assert sorted(assign_blocks([1.0] * 8, 8)) == list(range(8))
    mixed = [64.0] * 2 + [8.0] * 16 + [1.0] * 64                    # two large blocks, many small ones: 320 units over 8 ranks
    own = assign_blocks(mixed, 8)
    loads = [sum(c for c, o in zip(mixed, own) if o == r) for r in range(8)]
    assert max(loads) == 64.0 and min(loads) >= 30.0 and set(own) == set(range(8))
