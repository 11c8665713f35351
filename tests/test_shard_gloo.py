"""N>1 path on CPU: world_size-2 (and 3) `gloo` runs of the shard plan + gatherv/scatterv
(spz_amd/shard.py).  The per-rank fragments are produced by the CPU oracle here (test
infrastructure standing in for the GPU kernels, which need a GPU); what is under test is the
host logic: the plan's offsets, the grouped send/recv, the reassembled stream."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, deg, counts, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from oracle.pyoracle import Oracle
        from spz_amd import abi, shard
        from spz_amd.synth import FIELDS, floats_per_point, make_cloud_numpy

        O = Oracle()
        cloud = make_cloud_numpy(n, deg, 1234)  # every rank regenerates the same cloud from the seed
        plan = shard.plan_from_counts(counts, deg) if counts else shard.plan_even(n, deg, world)
        assert plan.num_points == n and sum(plan.count) == n
        a, c = plan.first[rank], plan.count[rank]
        sub = {k: cloud[k][a * floats_per_point(k, deg):(a + c) * floats_per_point(k, deg)] for k in FIELDS}
        local = torch.from_numpy(O.pack(sub, c, deg, True, 6))
        assert local.numel() == plan.local_layout(rank).total_bytes

        glob = None
        if rank == 0:
            glob = torch.zeros(plan.layout.total_bytes, dtype=torch.uint8)
            shard.write_global_header(glob, plan, antialiased=True)
        shard.gather_stream(local, plan, rank, glob)
        dist.barrier()
        ok = True
        if rank == 0:
            want = O.pack(cloud, n, deg, True, 6)
            ok = bool(np.array_equal(glob.numpy(), want))
            rc, hdr = abi.peek_header(glob.numpy().tobytes())
            ok = ok and rc == 0 and hdr.num_points == n
        # mirror image: scatter the global stream back, every rank must get its own bytes again
        back = torch.zeros_like(local)
        back[:16] = local[:16]
        shard.scatter_stream(glob, plan, rank, back)
        dist.barrier()
        ok = ok and bool(torch.equal(back, local))
        # async form
        if rank == 0:
            glob.zero_()
            shard.write_global_header(glob, plan, antialiased=True)
        works = shard.gather_stream(local, plan, rank, glob, async_op=True)
        for w in works:
            w.wait()
        dist.barrier()
        if rank == 0:
            ok = ok and bool(np.array_equal(glob.numpy(), O.pack(cloud, n, deg, True, 6)))
        q.put((rank, ok, ""))
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, False, traceback.format_exc() + repr(e)))


def _run(world, n, deg, counts=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, deg, counts, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, msg in results:
        assert ok, f"rank {rank} failed: {msg}"


def test_plan_even_covers_every_point_once():
    from spz_amd import shard
    for n in (0, 1, 15, 16, 17, 1000, 10_000_000):
        for world in (1, 2, 3, 8):
            p = shard.plan_even(n, 3, world)
            assert sum(p.count) == n
            assert all(p.first[r] + p.count[r] == (p.first[r + 1] if r + 1 < world else n) or p.count[r] == 0
                       for r in range(world))
            assert all(f % 16 == 0 or c == 0 for f, c in zip(p.first, p.count))
            g = p.layout
            covered = 0
            for r in range(world):
                for s, (goff, loff, nb) in enumerate(p.fragments(r)):
                    assert goff == g.offset[s] + p.first[r] * g.bytes_per_point[s]
                    assert loff == p.local_layout(r).offset[s]
                    covered += nb
            assert covered + 16 == g.total_bytes


def test_gatherv_two_ranks_gloo():
    _run(2, 5003, 3)


def test_gatherv_two_ranks_uneven_counts_gloo():
    _run(2, 1001, 1, counts=[1, 1000])


def test_gatherv_three_ranks_with_an_empty_shard_gloo():
    _run(3, 20, 2)  # plan_even(20, align 16) -> counts 16, 4, 0
