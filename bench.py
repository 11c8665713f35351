#!/usr/bin/env python3
"""bench.py — SPZ encode+decode throughput on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic input that is already
resident in HBM: encode (float SoA -> packed stream, `from` flip fused) followed by decode
(packed stream -> float SoA, `to` flip fused) of the same cloud.

  N = 1  workload = BASELINE configs[2]+[3]: 10 M synthetic Gaussians, SH degree 3, v3
         smallest-three rotations, encode from=RDF, decode to=RDF (RDF<->RUB fused).
  N > 1  weak scaling: every rank owns a 10 M-point shard of an N x 10 M-point stream
         (configs[4] at N = 8); per step each rank encodes its shard, ONE grouped gatherv
         (RCCL send/recv) reassembles the byte stream on rank 0, and each rank decodes the
         fragments it holds (floats stay sharded).  The gatherv overlaps the decode kernels.

Prints ONE JSON line on rank 0 (contract in the task statement), with
  roofline      : decode kernel, algorithmic bytes / HIP-event kernel time vs 8 TB/s HBM3E
  cpu_baseline  : the reference's own packGaussians+unpackGaussians (oracle/_ref, kind
                  "reference") or the C restatement (kind "port") on ONE host core.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
COORD = {"UNSPECIFIED": 0, "LDB": 1, "RDB": 2, "LUB": 3, "RUB": 4, "LDF": 5, "RDF": 6, "LUF": 7, "RUF": 8}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points", type=int, default=10_000_000, help="Gaussians per GPU")
    ap.add_argument("--sh-degree", type=int, default=3)
    ap.add_argument("--version", type=int, default=3)
    ap.add_argument("--from-coord", default="RDF")
    ap.add_argument("--to-coord", default="RDF")
    ap.add_argument("--no-collective", action="store_true", help="N>1: skip the gatherv (kernels only)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the measured configuration); gloo = rehearsal of the N>1 code path "
                         "on a box with fewer GPUs than ranks (host-staged exchange, ranks may share a GPU)")
    ap.add_argument("--route", default="auto", choices=["auto", "rccl", "torch", "ipc"],
                    help="N>1 exchange: rccl = spz_amd_gatherv_rccl (native: one ncclGroupStart/End, small sections sent "
                         "while sh is still encoding); torch = torch.distributed batch_isend_irecv; ipc = peers encode "
                         "straight into the root's stream over an IPC mapping.  auto = rccl on the nccl backend "
                         "(falls back to torch if the communicator cannot be made), torch on gloo")
    ap.add_argument("--placement", default="probe", choices=["probe", "plain"],
                    help="probe (default): the resident buffers come from spz_amd_cloud_buffers_alloc, which times a few "
                         "placements of the sh arrays and keeps the fast kind (DESIGN §10); plain: torch allocations as they fall")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-whole-file", action="store_true", help="skip the saveSpz / loadSpz figure (outside the timed region)")
    ap.add_argument("--cpu-sample-points", type=int, default=0, help="0 = the whole per-GPU workload")
    ap.add_argument("--traffic-file", default=os.path.join(ROOT, "profiles", "pmc_traffic.json"))
    return ap.parse_args()


KERNEL_SOURCES = ("spz_amd/csrc/spz_kernels.hip", "spz_amd/csrc/spz_kernel_params.hpp", "spz_amd/csrc/spz_abi.hip")


def kernel_source_sha256():
    """Hash of the sources that decide the kernels' memory traffic; profiles/pmc_traffic.json carries the hash
    of the tree its PMC passes ran on, so a traffic figure is only repeated for the kernels it was measured on."""
    import hashlib
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def algorithmic_bytes_per_point(sh_degree, version):
    """float bytes + packed bytes moved per Gaussian per direction."""
    d = {0: 0, 1: 9, 2: 24, 3: 45}[sh_degree]
    packed = (20 if version >= 3 else 19) + d
    return (14 + d) * 4 + packed


def cpu_shard_child(argv):
    """Child of cpu_all_cores: one host process, one shard, the reference's pack+unpack; never touches the GPU."""
    per, deg, frm, to, seed, tmp, idx = int(argv[0]), int(argv[1]), int(argv[2]), int(argv[3]), int(argv[4]), argv[5], argv[6]
    from oracle.pyoracle import Reference
    from spz_amd.synth import make_cloud_numpy
    R = Reference()
    c = make_cloud_numpy(per, deg, seed)
    R.bench_pack_unpack(c, min(per, 1000), deg, frm, to)          # page the library in
    open(os.path.join(tmp, f"ready.{idx}"), "w").close()
    go = os.path.join(tmp, "go")
    deadline = time.monotonic() + 300
    while not os.path.exists(go):
        if time.monotonic() > deadline:
            return 3
        time.sleep(0.002)
    t0 = time.perf_counter()
    tp, tu, _ = R.bench_pack_unpack(c, per, deg, frm, to)
    print(json.dumps({"wall": time.perf_counter() - t0, "pack": tp, "unpack": tu}), flush=True)
    return 0


def cpu_all_cores(m, sh_degree, frm, to):
    import shutil
    import tempfile
    procs = max(1, min(16, os.cpu_count() or 1))
    per = m // procs
    if procs < 2 or per < 1000:
        return None
    tmp = tempfile.mkdtemp(prefix="spz_cpu_shards_")
    kids = []
    try:
        for t in range(procs):
            kids.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-shard-child", str(per),
                                          str(sh_degree), str(frm), str(to), str(1000 + t), tmp, str(t)],
                                         stdout=subprocess.PIPE, text=True, cwd=ROOT,
                                         env=dict(os.environ, SPZ_AMD_NO_HIP_PRELOAD="1")))
        deadline = time.monotonic() + 300
        while sum(os.path.exists(os.path.join(tmp, f"ready.{t}")) for t in range(procs)) < procs:
            if time.monotonic() > deadline or any(k.poll() is not None for k in kids):
                raise RuntimeError("a shard process did not get ready")
            time.sleep(0.01)
        open(os.path.join(tmp, "go"), "w").close()
        walls = []
        for k in kids:
            out, _ = k.communicate(timeout=600)
            if k.returncode != 0:
                raise RuntimeError("a shard process failed")
            walls.append(json.loads(out.strip().splitlines()[-1])["wall"])
        return {"value": per * procs / max(walls), "unit": "Gaussians/s", "cores": procs,
                "sample": f"{procs} processes x {per} Gaussians each (same distributions, seeds 1000..), pack+unpack, "
                          f"slowest {max(walls):.2f} s, fastest {min(walls):.2f} s"}
    except (OSError, RuntimeError, ValueError, subprocess.SubprocessError) as e:
        return {"value": None, "error": str(e)}
    finally:
        for k in kids:
            if k.poll() is None:
                k.kill()
        shutil.rmtree(tmp, ignore_errors=True)


def cpu_baseline(cloud_t, n, sh_degree, frm, to, sample_points, gpu_stream_fn, gpu_decoded_bit_sums_fn=None):
    """Times the reference C++ (or the C port) on one host core over the same synthetic points and
    checks that what was timed produced the same bytes as the GPU path."""
    import numpy as np
    from oracle.pyoracle import REF_SO, Oracle, Reference
    from spz_amd.synth import FIELDS, floats_per_point

    m = n if sample_points <= 0 else min(n, sample_points)
    host = {k: cloud_t[k][: m * floats_per_point(k, sh_degree)].cpu().numpy() for k in FIELDS}
    gpu_stream = gpu_stream_fn(m)
    decoded_identical = None
    if os.path.exists(REF_SO):
        kind = "reference"
        R = Reference()
        t_pack, t_unpack, stream = R.bench_pack_unpack(host, m, sh_degree, frm, to, want_stream=True)
        if gpu_decoded_bit_sums_fn is not None:
            decoded_identical = bool(R.last_decoded_bit_sums == gpu_decoded_bit_sums_fn(m))
    else:
        kind = "port"
        O = Oracle()
        t0 = time.perf_counter()
        stream = O.pack(host, m, sh_degree, False, frm)
        t1 = time.perf_counter()
        O.unpack(stream, to)
        t2 = time.perf_counter()
        t_pack, t_unpack = t1 - t0, t2 - t1
    parity = bool(stream.size == gpu_stream.size and np.array_equal(stream, gpu_stream))
    # the "all cores" figure — the same reference code over P independent point-range shards
    # in P processes at once (no code change to it), P = this box's CPU share for one GPU.
    all_cores = cpu_all_cores(m, sh_degree, frm, to) if kind == "reference" else None
    return {
        "value": m / (t_pack + t_unpack),
        "unit": "Gaussians/s",
        "cores": 1,
        "kind": kind,
        "sample": f"{m} of the {n} Gaussians of the N=1 workload (SH{sh_degree}), packGaussians "
                  f"{t_pack:.2f} s + unpackGaussians {t_unpack:.2f} s, gzip excluded; single-threaded as shipped, "
                  f"host has {os.cpu_count()} cores",
        "pack_gaussians_per_s": m / t_pack,
        "unpack_gaussians_per_s": m / t_unpack,
        "stream_bit_identical_to_gpu": parity,
        "decoded_bit_sums_identical_to_gpu": decoded_identical,
        "all_cores": all_cores,
    }


def whole_file(cloud, n, deg, frm, to):
    """Outside the timed region, N = 1: the reference's own entry points on the same cloud with host vectors in
    and out — spz::saveSpz (pack over PCIe + the gzip container, the reference's bytes) and spz::loadSpz of what it
    wrote — timed around the C++ calls (spz_amd.spz._save_load_seconds).  Never part of `value`."""
    try:
        import spz_amd.spz as spz
        from spz_amd.synth import FIELDS
        g = spz.GaussianCloud()
        g.sh_degree = deg
        for k in FIELDS:
            setattr(g, k, cloud[k].cpu().numpy())
        po, uo = spz.PackOptions(), spz.UnpackOptions()
        po.from_coord = spz.CoordinateSystem(frm)
        uo.to_coord = spz.CoordinateSystem(to)
        best = None
        for _ in range(3):
            save_s, load_s, nbytes, back, on_device = spz._save_load_seconds(g, po, uo)
            best = (save_s, load_s, nbytes, back, on_device) if best is None else (
                min(best[0], save_s), min(best[1], load_s), nbytes, back, on_device)
        return {"save_spz_s": round(best[0], 4), "load_spz_s": round(best[1], 4), "spz_bytes": best[2],
                "points_read_back": best[3], "gzip_stage_on_device": bool(best[4]),
                "note": "spz::saveSpz / spz::loadSpz (vector overloads) of the N=1 cloud, host vectors in and out, each the best of 3; "
                        "the member is byte-identical to zlib's (tests/test_gpu_gzip_device.py)"}
    except Exception as e:  # the figure is an extra: the bench line does not depend on it
        return {"error": f"{type(e).__name__}: {e}"}


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-shard-child":
        sys.exit(cpu_shard_child(sys.argv[2:]))
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        # Launched directly with --gpus N: start the ranks as child processes (never exec after
        # the GPU may have been initialised) and leave with their exit code.
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this pool
    import torch
    import torch.distributed as dist

    from spz_amd import abi, device as D, shard
    from spz_amd.synth import FIELDS, floats_per_point, make_cloud_torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    if args.backend == "gloo":
        local_rank %= torch.cuda.device_count()   # rehearsal: ranks may share a card
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        from datetime import timedelta
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=timedelta(seconds=300))
        else:
            dist.init_process_group("gloo", timeout=timedelta(seconds=300))

    n, deg, ver = args.points, args.sh_degree, args.version
    frm, to = COORD[args.from_coord], COORD[args.to_coord]
    lay = abi.stream_layout(n, deg, ver)
    hdr = D.make_header(n, deg, ver)
    use_coll = distributed and not args.no_collective
    plan = shard.plan_from_counts([n] * world, deg, ver) if distributed else None
    # With the collective, stream buffers are double-buffered so that the gatherv of step k (link-bound,
    # on RCCL's own stream) overlaps the encode and decode kernels of steps k+1 and k+2.
    nbuf = 2 if use_coll else 1
    placement = {"method": "plain", "note": "torch allocations as they fall: which placement kind the sh arrays get is luck (DESIGN §10)"}
    placed = []
    if args.placement == "probe" and not use_coll:
        # Where the sh arrays lie relative to the other buffers of a launch decides +-10 % of both kernels (DESIGN §10,
        # profiles/r03_placement_*): the library makes the resident buffers and keeps, of a few placements it times with
        # the launch itself, one of the fast kind.  Done once, before the data exists; not part of any timed region.
        p_out = D.alloc_placed(n, deg, dev, ver, None, "decode")
        p_in = D.alloc_placed(n, deg, dev, ver, p_out.stream, "encode")
        placed = [p_out, p_in]
        streams = [p_out.stream[:lay.total_bytes]]
        out = p_out.cloud
        cloud = p_in.cloud
        gen = make_cloud_torch(n, deg, 3 + 47 * rank, dev)           # seeds 3, 50, 97, ... per rank
        for k in FIELDS:
            cloud[k].copy_(gen[k])
        del gen
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        placement = {"method": "probe", "decode_buffers": p_out.report, "encode_buffers": p_in.report,
                     "note": "spz_amd_cloud_buffers_alloc: the sh array in an allocation of its own, chosen among up to 6 by timing "
                             "the launch on zeroed buffers before the data was made; outside every timed region"}
    else:
        cloud = make_cloud_torch(n, deg, 3 + 47 * rank, dev)           # seeds 3, 50, 97, ... per rank
        out = D.alloc_cloud(n, deg, dev)
        streams = [torch.empty(lay.total_bytes, dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    stream = streams[0]
    global_streams = [None] * nbuf
    route = None
    if use_coll:
        route = args.route if args.route != "auto" else ("rccl" if args.backend == "nccl" else "torch")
    if use_coll and rank == 0 and route != "ipc":
        for b in range(nbuf):
            global_streams[b] = torch.empty(plan.layout.total_bytes, dtype=torch.uint8, device=dev)
            shard.write_global_header(global_streams[b], plan)

    K, W = args.steps, args.warmup
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(K)]

    # ---- the exchange route ----------------------------------------------------------------------------
    route_note = None
    gx, comm_stream, ipcs = None, None, None
    if use_coll:
        if route == "rccl":
            # agree first, then build: a rank without RCCL must not leave the others inside ncclCommInitRank
            flag_dev = dev if args.backend == "nccl" else "cpu"
            can = torch.tensor([int(abi.load_library().spz_amd_rccl_available())], dtype=torch.int32, device=flag_dev)
            dist.all_reduce(can, op=dist.ReduceOp.MIN)
            err = None if int(can.item()) == 1 else "librccl.so.1 is not loadable on every rank"
            if err is None:
                try:
                    gx = shard.RcclGather(plan, rank)
                except (RuntimeError, OSError) as e:   # communicator refused (e.g. ranks sharing a GPU)
                    err = str(e)
                ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=flag_dev)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok.item()) == 0:
                    err = err or "another rank could not create its communicator"
            if err is not None:
                if gx is not None:
                    gx.close()
                gx, route = None, "torch"
                route_note = f"native RCCL route unavailable ({err}): fell back to torch.distributed"
            else:
                comm_stream = torch.cuda.Stream(device=dev)
        if route == "ipc":
            ipcs = [shard.IpcGlobalStream(plan, rank) for _ in range(nbuf)]
            if rank == 0:
                for b in range(nbuf):
                    shard.write_global_header(ipcs[b].raw.tensor(dev), plan)
                    global_streams[b] = ipcs[b].raw.tensor(dev)
    ghdr = D.make_header(plan.num_points, deg, ver) if use_coll else None
    ev_small = [torch.cuda.Event() for _ in range(nbuf)]
    ev_sh = [torch.cuda.Event() for _ in range(nbuf)]
    done_ev = [None] * nbuf

    if route == "ipc":
        # every rank encodes its shard straight into the root's stream (the kernel's stores ARE the exchange) and
        # decodes its fragments from there; a peer's decode therefore reads them back over xGMI
        views = [ipcs[b].raw.tensor(dev) for b in range(nbuf)]
        run_encode = lambda b: D.encode_shard(cloud, plan.first[rank], n, plan.num_points, deg, ipcs[b].raw,
                                              from_coord=frm, version=ver, write_header=(rank == 0))
        run_decode = lambda b: D.decode_shard(views[b], ghdr, plan.first[rank], n, to, out=out)
    elif use_coll and rank == 0:
        # the root encodes straight into (and decodes straight from) its slot of the global stream
        run_encode = lambda b: D.encode_shard(cloud, plan.first[0], n, plan.num_points, deg, global_streams[b],
                                              from_coord=frm, version=ver, write_header=True)
        run_decode = lambda b: D.decode_shard(global_streams[b], ghdr, plan.first[0], n, to, out=out)
    elif route == "rccl":
        def run_encode(b):
            # the five small sections first (20 B/point): their fragments travel while the sh section (45 B/point) encodes
            D.encode_shard(cloud, 0, n, n, deg, streams[b], from_coord=frm, version=ver, write_header=True,
                           section_mask=abi.SMALL_SECTIONS)
            ev_small[b].record()
            D.encode_shard(cloud, 0, n, n, deg, streams[b], from_coord=frm, version=ver, section_mask=abi.SH_SECTION)
            ev_sh[b].record()
        run_decode = lambda b: D.decode(streams[b], hdr, to, out=out)
    else:
        run_encode = lambda b: D.encode(cloud, n, deg, False, frm, ver, out=streams[b])
        run_decode = lambda b: D.decode(streams[b], hdr, to, out=out)

    pending = [[] for _ in range(nbuf)]
    counter = [0]

    def drain(b):
        for w in pending[b]:
            w.wait()       # RCCL: the current stream waits for the gatherv that last used buffer b
        pending[b] = []
        if done_ev[b] is not None:
            torch.cuda.current_stream().wait_event(done_ev[b])
            done_ev[b] = None

    def exchange(b):
        if route == "torch":
            pending[b] = shard.gather_stream(None if rank == 0 else streams[b], plan, rank, global_streams[b],
                                             async_op=True)
        elif route == "rccl":
            local = None if rank == 0 else streams[b]
            if rank != 0:
                comm_stream.wait_event(ev_small[b])
            gx.gather(local, global_streams[b], abi.SMALL_SECTIONS, stream=comm_stream)
            if rank != 0:
                comm_stream.wait_event(ev_sh[b])
            gx.gather(local, global_streams[b], abi.SH_SECTION, stream=comm_stream)
            done_ev[b] = torch.cuda.Event()
            done_ev[b].record(comm_stream)

    def step(k, timed):
        b = counter[0] % nbuf
        counter[0] += 1
        drain(b)
        e = ev[k] if timed else None
        if e: e[0].record()
        run_encode(b)
        if e: e[1].record()
        if use_coll:
            exchange(b)
        if e: e[2].record()
        run_decode(b)
        if e: e[3].record()
        return b

    def fence():
        for b in range(nbuf):
            drain(b)
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(W):
        step(k, False)
    fence()

    # N>1 parity gate: every fragment that landed in the root's global stream must be the bytes its
    # rank encoded (per-section byte sums, exchanged with one small all_gather outside the timed region).
    gather_verified = None
    if use_coll:
        last = step(0, False)     # one more untimed step whose buffers are then checked
        fence()
        global_stream = global_streams[last]
        if route == "ipc" and rank != 0:   # what this rank's bytes must be: its shard encoded once more, locally
            D.encode(cloud, n, deg, False, frm, ver, out=streams[last])
            torch.cuda.synchronize()
        src_buf = global_stream if rank == 0 else streams[last]
        mine = torch.stack([src_buf[(g if rank == 0 else l):(g if rank == 0 else l) + nb].sum(dtype=torch.int64)
                            for g, l, nb in plan.fragments(rank)])
        if args.backend != "nccl":
            mine = mine.cpu()
        sums = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(sums, mine)
        if rank == 0:
            gather_verified = True
            for r in range(world):
                got = torch.stack([global_stream[g:g + nb].sum(dtype=torch.int64) for g, _, nb in plan.fragments(r)])
                gather_verified = gather_verified and bool(torch.equal(got.cpu(), sums[r].cpu()))
    # a named range around the timed steps: under `rocprofv3 --marker-trace --kernel-trace` tools/summarize_profile.py then
    # averages exactly these launches (the placement probe, the warm-up and the checks after launch the same kernels)
    try:
        torch.cuda.nvtx.range_push("spz_bench_timed_region")
        marked = True
    except Exception:  # noqa: BLE001  (no roctx in this process: the summary then covers every launch)
        marked = False
    t0 = time.perf_counter()
    for k in range(K):
        step(k, True)
    fence()
    elapsed = time.perf_counter() - t0
    if marked:
        torch.cuda.nvtx.range_pop()
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # N>1: the same K steps once more WITHOUT the exchange (every rank encodes and decodes its own shard,
    # nothing crosses xGMI), so the line shows what the kernels sustain next to the link-bound headline.
    shards_only = None
    if use_coll:
        t1 = time.perf_counter()
        for k in range(K):
            D.encode(cloud, n, deg, False, frm, ver, out=streams[0])
            D.decode(streams[0], hdr, to, out=out)
        fence()
        el2 = time.perf_counter() - t1
        t = torch.tensor([el2], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el2 = float(t.item())
        into_root = sum(nb for r in range(1, world) for _, _, nb in plan.fragments(r))
        shards_only = {
            "value": n * world * K / el2, "unit": "Gaussians/s", "ms_per_step": el2 / K * 1e3,
            "note": "same shards and kernels, no gatherv: the data-parallel rate of the path itself",
            "gatherv_bytes_into_root_per_step": into_root,
            "gatherv_GBps_into_root": into_root * K / elapsed / 1e9,
            "gatherv_GBps_per_peer_link": into_root * K / elapsed / 1e9 / (world - 1),
        }

    enc_ms = sum(e[0].elapsed_time(e[1]) for e in ev) / K
    dec_ms = sum(e[2].elapsed_time(e[3]) for e in ev) / K

    # sanity inside the bench: the stream decodes back to what a re-encode reproduces
    fixed_point = None
    if frm == to and not (use_coll and rank == 0):
        # quantisation is idempotent except for smallest-three near-ties (format property): every
        # section but the rotations must re-encode to identical bytes
        stream = streams[(counter[0] - 1) % nbuf]
        s2 = D.encode(out, n, deg, False, to, ver)
        torch.cuda.synchronize()
        o_rot = lay.offset[abi.SEC_ROTATIONS]
        e_rot = o_rot + lay.bytes[abi.SEC_ROTATIONS]
        rot_w = lay.bytes_per_point[abi.SEC_ROTATIONS]
        fixed_point = {
            "non_rotation_sections_identical": bool(torch.equal(s2[:o_rot], stream[:o_rot]) and
                                                    torch.equal(s2[e_rot:], stream[e_rot:])),
            "rotations_changed": int((s2[o_rot:e_rot].reshape(-1, rot_w) !=
                                      stream[o_rot:e_rot].reshape(-1, rot_w)).any(dim=1).sum()),
        }

    # BASELINE config 4, outside the timed region: the reference-shaped decode (decode to RUB, then a
    # separate convertCoordinates pass, load-spz.cc:529) next to the fused decode used above.
    two_pass = None
    if world == 1 and to != 0:
        e2 = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        reps = 5
        stream = streams[(counter[0] - 1) % nbuf]
        D.decode(stream, hdr, 0, out=out)
        D.convert_coordinates(out, n, deg, 4, to)
        torch.cuda.synchronize()
        e2[0].record()
        for _ in range(reps):
            D.decode(stream, hdr, 0, out=out)
            D.convert_coordinates(out, n, deg, 4, to)
        e2[1].record()
        torch.cuda.synchronize()
        two_pass = {"fused_decode_ms": dec_ms, "decode_then_flip_pass_ms": e2[0].elapsed_time(e2[1]) / reps,
                    "algorithmic_bytes_two_pass": n * (algorithmic_bytes_per_point(deg, ver) + 2 * 4 * (3 + 4 + {0: 0, 1: 9, 2: 24, 3: 45}[deg]))}
        run_decode((counter[0] - 1) % nbuf)  # leave `out` as the fused decode produced it
        torch.cuda.synchronize()

    if rank == 0:
        bpp = algorithmic_bytes_per_point(deg, ver)
        total_points = n * world
        # roofline.traffic is NOT measured by this run (PMC counters need rocprofv3 around the process): it is the
        # figure of the committed PMC passes, repeated only when they ran on these same kernel sources and workload
        traffic, traffic_source = None, {"measured_in_this_run": False, "file": os.path.relpath(args.traffic_file, ROOT)}
        if os.path.exists(args.traffic_file):
            try:
                with open(args.traffic_file) as f:
                    tj = json.load(f)
                traffic_source.update({"tag": tj.get("tag"), "passes": tj.get("source"),
                                       "kernel_source_sha256": tj.get("kernel_source_sha256")})
                same_workload = tj.get("points") == n and tj.get("sh_degree") == deg
                same_kernels = tj.get("kernel_source_sha256") == kernel_source_sha256()
                traffic_source["matches_this_workload"] = same_workload
                traffic_source["matches_current_kernel_sources"] = same_kernels
                if same_workload and same_kernels:
                    traffic = tj.get("decode_hbm_bytes_per_launch")
            except (OSError, ValueError):
                traffic = None
        res = {
            "metric": "Gaussians/s encode+decode (SH3, 10M pts)",
            "value": total_points * K / elapsed,
            "unit": "Gaussians/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{n} synthetic Gaussians per GPU, SH degree {deg}, v{ver} "
                            f"({'smallest-three' if ver >= 3 else 'first-three'} rotations), encode from={args.from_coord} "
                            f"+ decode to={args.to_coord} with the coordinate flips fused, inputs resident in HBM",
                "points_per_gpu": n, "sh_degree": deg, "version": ver,
                "parallelism": ("single GPU" if world == 1 else
                                f"point-range shards x{world}" + ({
                                    "rccl": ", gatherv of the byte stream to rank 0 by spz_amd_gatherv_rccl (one ncclGroupStart/End "
                                            "per section group, the small sections sent while sh encodes), double-buffered",
                                    "torch": ", one grouped gatherv of the byte stream to rank 0 per step "
                                             "(torch.distributed batch_isend_irecv), double-buffered",
                                    "ipc": ", peers encode straight into rank 0's stream over an IPC mapping (no second pass)",
                                }[route] if use_coll else ", no collective") +
                                ("" if args.backend == "nccl" else " [REHEARSAL: gloo backend, host-staged, not a measurement]")),
            },
            "roofline": {
                "bound": "hbm", "kernel": "spz_decode_kernel",
                "achieved": n * bpp / (dec_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": n * bpp / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "traffic": traffic, "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": n * bpp, "avg_launch_ms": dec_ms,
            },
            "roofline_encode": {
                "bound": "hbm", "kernel": "spz_encode_kernel",
                "achieved": n * bpp / (enc_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": n * bpp / (enc_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "algorithmic_bytes_per_launch": n * bpp, "avg_launch_ms": enc_ms,
            },
            "encode_gaussians_per_s_per_gpu": n / (enc_ms * 1e-3),
            "decode_gaussians_per_s_per_gpu": n / (dec_ms * 1e-3),
            "reencode_fixed_point": fixed_point,
            "gather_verified": gather_verified,
            "exchange_route": route, "exchange_route_note": route_note,
            "shards_only": shards_only,
            "config4_fused_vs_two_pass": two_pass,
            "placement": placement,
        }
        if world == 1 and not args.no_cpu_baseline:
            def gpu_stream_fn(m):
                sub = {k: cloud[k][: m * floats_per_point(k, deg)] for k in FIELDS}
                s = D.encode(sub, m, deg, False, frm, 3)
                torch.cuda.synchronize()
                return s.cpu().numpy()
            def gpu_decoded_bit_sums_fn(m):
                # per-array sum of the decoded floats' bit patterns (what the reference shim sums on its side)
                sub = {k: cloud[k][: m * floats_per_point(k, deg)] for k in FIELDS}
                s = D.encode(sub, m, deg, False, frm, 3)
                d = D.decode(s, D.make_header(m, deg, 3), to)
                torch.cuda.synchronize()
                sums = []
                for k in FIELDS:
                    v = d[k].view(torch.int32)
                    total = 0
                    for part in v.split(1 << 26):   # bounded temporaries
                        total += int((part.to(torch.int64) & 0xffffffff).sum().item())
                    sums.append(total & 0xffffffffffffffff)
                return sums
            res["cpu_baseline"] = cpu_baseline(cloud, n, deg, frm, to, args.cpu_sample_points, gpu_stream_fn,
                                               gpu_decoded_bit_sums_fn)
        if world == 1 and not args.no_whole_file:
            res["whole_file"] = whole_file(cloud, n, deg, frm, to)
        print(json.dumps(res), flush=True)
    if distributed:
        torch.cuda.synchronize()
        dist.barrier()
        if gx is not None:
            gx.close()
        if ipcs is not None:
            if rank == 0:
                global_streams[:] = [None] * nbuf
            views = None
            dist.barrier()               # no peer may still have the root's buffer in use
            for i in ipcs:
                if rank != 0:
                    i.close()
            dist.barrier()
            for i in ipcs:
                if rank == 0:
                    i.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
