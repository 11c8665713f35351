"""Point-range sharding of one SPZ stream across ranks, and the gatherv that reassembles it.

Gaussians are independent, so a cloud of N points is split into contiguous point ranges, one
per rank (one process per GPU).  The stream is attribute-major (load-spz.cc:540-545), so rank
r's output is six fragments, and section s of the final stream is the concatenation over
ranks of fragment (r, s).  The only exchange step on the path is therefore a gatherv of
6 x (R-1) byte ranges into their final offsets on the root: one grouped batch of point-to-point
sends/receives (RCCL has no native gatherv; `torch.distributed.batch_isend_irecv` wraps the
batch in ncclGroupStart/End on the nccl(=RCCL) backend, and works unchanged over gloo on CPU
tensors, which is how the tests cover it).  Over xGMI every peer has its own link to the root,
so the floor is max_r(fragment bytes of r) / link rate, not a ring.

The decode direction needs no exchange when the floats stay sharded: every rank decodes the
fragments it already holds (`scatter_stream` exists for the case where the stream starts on
the root only).
"""
from dataclasses import dataclass
from typing import List

import torch
import torch.distributed as dist

from . import abi

SECTIONS = 6


@dataclass(frozen=True)
class ShardPlan:
    """Where every rank's fragments live in the global stream."""
    num_points: int
    sh_degree: int
    version: int
    world_size: int
    first: List[int]   # first point of rank r
    count: List[int]   # number of points of rank r

    @property
    def layout(self):
        return abi.stream_layout(self.num_points, self.sh_degree, self.version)

    def local_layout(self, rank):
        return abi.stream_layout(self.count[rank], self.sh_degree, self.version)

    def fragments(self, rank):
        """[(global_offset, local_offset, nbytes)] for the six sections of `rank`."""
        g, l = self.layout, self.local_layout(rank)
        out = []
        for s in range(SECTIONS):
            bpp = g.bytes_per_point[s]
            out.append((g.offset[s] + self.first[rank] * bpp, l.offset[s], self.count[rank] * bpp))
        return out


def plan_even(num_points, sh_degree, world_size, version=3, align=16):
    """Contiguous ranges of ceil(N/R) points rounded up to `align` points, so that every fragment
    starts 16-byte aligned inside its section; the last ranks may be short or empty."""
    per = -(-num_points // world_size)
    per = -(-per // align) * align
    first, count = [], []
    for r in range(world_size):
        a = min(num_points, r * per)
        b = min(num_points, (r + 1) * per)
        first.append(a)
        count.append(b - a)
    return ShardPlan(num_points, sh_degree, version, world_size, first, count)


def plan_from_counts(counts, sh_degree, version=3):
    """Weak-scaling plan: rank r owns counts[r] points; the global stream holds their sum."""
    first, acc = [], 0
    for c in counts:
        first.append(acc)
        acc += c
    return ShardPlan(acc, sh_degree, version, len(counts), first, list(counts))


def _needs_host_staging(t, group):
    """gloo has no CUDA point-to-point: with that backend CUDA fragments go through host copies
    (used to rehearse the N>1 path on a box without RCCL peers; RCCL sends device memory directly)."""
    return t is not None and t.is_cuda and dist.get_backend(group) == "gloo"


class _StagedWork:
    """Work handle of a host-staged receive: wait() finishes the transfer, then copies to the device."""

    def __init__(self, works, copies):
        self.works, self.copies = works, copies

    def wait(self):
        for w in self.works:
            w.wait()
        for dst, src in self.copies:
            dst.copy_(src)


def gather_stream(local_stream, plan, rank, global_stream=None, dst=0, group=None, async_op=False):
    """Gatherv of the ranks' fragments into the root's global stream.

    local_stream : uint8 tensor, this rank's own stream (header + its six fragments, as produced
                   by encoding its `count` points on their own).  The root may pass None if it
                   encoded straight into `global_stream` with encode_shard.
    global_stream: root only, uint8 tensor of plan.layout.total_bytes; fragments land at their
                   final offsets, the header is written by the root.
    Returns the list of outstanding work handles when async_op, else waits for them.
    """
    ops, copies = [], []
    staged = _needs_host_staging(global_stream if rank == dst else local_stream, group)
    if rank == dst:
        assert global_stream is not None and global_stream.numel() >= plan.layout.total_bytes
        for r in range(plan.world_size):
            if r == dst:
                if local_stream is not None:
                    for goff, loff, nb in plan.fragments(r):
                        if nb:
                            global_stream[goff:goff + nb].copy_(local_stream[loff:loff + nb], non_blocking=True)
                continue
            for goff, _, nb in plan.fragments(r):
                if nb:
                    target = global_stream[goff:goff + nb]
                    if staged:
                        host = torch.empty(nb, dtype=torch.uint8)
                        copies.append((target, host))
                        target = host
                    ops.append(dist.P2POp(dist.irecv, target, r, group))
    else:
        for _, loff, nb in plan.fragments(rank):
            if nb:
                src = local_stream[loff:loff + nb]
                ops.append(dist.P2POp(dist.isend, src.cpu() if staged else src, dst, group))
    works = dist.batch_isend_irecv(ops) if ops else []
    if staged and works:
        works = [_StagedWork(works, copies)]
    if async_op:
        return works
    for w in works:
        w.wait()
    return []


def scatter_stream(global_stream, plan, rank, local_stream, src=0, group=None, async_op=False):
    """Mirror image: the root sends every rank its six fragments (65 B/point for SH3)."""
    ops = []
    if rank == src:
        for r in range(plan.world_size):
            for goff, loff, nb in plan.fragments(r):
                if not nb:
                    continue
                if r == src:
                    local_stream[loff:loff + nb].copy_(global_stream[goff:goff + nb], non_blocking=True)
                else:
                    ops.append(dist.P2POp(dist.isend, global_stream[goff:goff + nb], r, group))
    else:
        for _, loff, nb in plan.fragments(rank):
            if nb:
                ops.append(dist.P2POp(dist.irecv, local_stream[loff:loff + nb], src, group))
    works = dist.batch_isend_irecv(ops) if ops else []
    if async_op:
        return works
    for w in works:
        w.wait()
    return []


class RcclGather:
    """The RCCL route through the C ABI (spz_amd_gatherv_rccl): one ncclGroupStart/End of send/recv pairs that
    land every fragment at its final offset on the root.  The communicator is this class's own
    (ncclCommInitRank on a unique id that rank `dst` creates and torch.distributed — any backend — hands round);
    one rank per GPU, as RCCL requires."""

    def __init__(self, plan, rank, dst=0, group=None):
        import ctypes as C
        self.plan, self.rank, self.dst = plan, rank, dst
        self.L = abi.load_library()
        self.comm = None
        # every rank takes part in the broadcast whatever happened before it, so that a rank that cannot load RCCL
        # (or a root that cannot make the id) makes ALL ranks raise instead of leaving the others waiting
        ident = [None]
        if rank == dst and self.L.spz_amd_rccl_available():
            buf = (C.c_uint8 * abi.RCCL_UNIQUE_ID_BYTES)()
            if self.L.spz_amd_rccl_unique_id(buf) == abi.OK:
                ident = [bytes(buf)]
        dist.broadcast_object_list(ident, src=dst, group=group)
        if ident[0] is None:
            raise RuntimeError("the root rank could not create an RCCL unique id (librccl.so.1 not loadable there?)")
        if not self.L.spz_amd_rccl_available():
            raise RuntimeError("librccl.so.1 cannot be loaded on this rank")
        self.comm = C.c_void_p()
        buf = (C.c_uint8 * abi.RCCL_UNIQUE_ID_BYTES).from_buffer_copy(ident[0])
        abi.check(self.L.spz_amd_rccl_comm_init(buf, plan.world_size, rank, C.byref(self.comm)), "spz_amd_rccl_comm_init")
        self.first = (C.c_uint64 * plan.world_size)(*plan.first)
        self.count = (C.c_uint64 * plan.world_size)(*plan.count)

    def gather(self, local_stream, global_stream, section_mask=abi.ALL_SECTIONS, stream=None):
        """Enqueue on `stream` (torch.cuda.Stream; None = current).  local_stream: this rank's own stream tensor
        (None on a root that encoded straight into global_stream); global_stream: root only."""
        s = (stream or torch.cuda.current_stream()).cuda_stream
        lp = local_stream.data_ptr() if local_stream is not None else None
        gp = None
        if global_stream is not None:
            gp = global_stream.ptr if hasattr(global_stream, "ptr") else global_stream.data_ptr()
        rc = self.L.spz_amd_gatherv_rccl(self.comm, self.rank, self.plan.world_size, self.dst, self.first, self.count,
                                         self.plan.sh_degree, self.plan.version, lp, gp, int(section_mask), s)
        if rc == abi.ERR_COMM:
            raise abi.SpzAmdError(rc, f"spz_amd_gatherv_rccl (ncclResult {self.L.spz_amd_last_rccl_error()})")
        abi.check(rc, "spz_amd_gatherv_rccl")

    def close(self):
        if self.comm:
            self.L.spz_amd_rccl_comm_destroy(self.comm)
            self.comm = None


class IpcGlobalStream:
    """The IPC route: rank `dst` allocates the global stream and exports it; every other rank of the node maps
    it and encodes its shard straight into it (device.encode_shard(..., out=self.raw)), so the encode
    kernel's stores are the exchange.  `raw` is a device.RawStream on every rank."""

    def __init__(self, plan, rank, dst=0, group=None):
        import ctypes as C
        from .device import RawStream
        self.L = abi.load_library()
        self.owner = rank == dst
        nbytes = plan.layout.total_bytes
        ptr = C.c_void_p()
        handle = [None]
        if self.owner:
            buf = (C.c_uint8 * abi.IPC_HANDLE_BYTES)()
            abi.check(self.L.spz_amd_ipc_alloc(nbytes, C.byref(ptr), buf), "spz_amd_ipc_alloc")
            handle = [bytes(buf)]
        dist.broadcast_object_list(handle, src=dst, group=group)
        if not self.owner:
            buf = (C.c_uint8 * abi.IPC_HANDLE_BYTES).from_buffer_copy(handle[0])
            abi.check(self.L.spz_amd_ipc_open(buf, C.byref(ptr)), "spz_amd_ipc_open")
        self.raw = RawStream(ptr.value, nbytes)

    def close(self):
        if self.raw is not None:
            (self.L.spz_amd_ipc_free if self.owner else self.L.spz_amd_ipc_close)(self.raw.ptr)
            self.raw = None


def write_global_header(global_stream, plan, antialiased=False):
    hdr = abi.write_header(plan.version, plan.num_points, plan.sh_degree, 12, antialiased)
    global_stream[:16].copy_(torch.frombuffer(bytearray(hdr), dtype=torch.uint8))
    return global_stream
