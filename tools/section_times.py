import ctypes as C, json, sys, os, statistics
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from spz_amd import abi, device as D
from spz_amd.synth import FIELDS, make_cloud_torch
dev = torch.device("cuda:0")
L = abi.load_library()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for n, deg, ver in ((1_000_000, 0, 2), (1_000_000, 0, 3), (10_000_000, 0, 3)):
    cloud = make_cloud_torch(n, deg, 3, dev)
    lay = abi.stream_layout(n, deg, ver)
    stream = torch.empty(lay.total_bytes, dtype=torch.uint8, device=dev)
    pin = abi.CloudPtrs(*[cloud[k].data_ptr() for k in FIELDS])
    names = ["positions", "alphas", "colors", "scales", "rotations", "sh"]
    res = {"points": n, "sh_degree": deg, "version": ver}
    for label, mask in [("all", 0x3f)] + [(names[i], 1 << i) for i in range(5)] + [("all_but_rot", 0x2f), ("pos+rot", 0x11)]:
        def run():
            rc = L.spz_amd_encode_shard_sections_device(C.byref(pin), 0, n, n, deg, 0, 6, ver, 1, mask, stream.data_ptr(), stream.numel(), s)
            assert rc == 0, rc
        for _ in range(5): run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(15):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            e[0].record()
            for _ in range(20): run()
            e[1].record(); torch.cuda.synchronize()
            ts.append(e[0].elapsed_time(e[1]) / 20 * 1e3)
        res[label + "_us"] = round(statistics.median(ts), 2)
    print(json.dumps(res), flush=True)
