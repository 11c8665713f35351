#!/bin/bash
# round 3: gzip tests on the vector-load bit packer, laps, then ten bench lines from fresh processes (placement probe)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_gzip_device.py tests/test_gpu_container_safety.py tests/test_gpu_placement.py -x -q > $O/pytest_s11.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -n 6 $O/pytest_s11.log
[ $rc -eq 0 ] || exit $rc
SPZ_AMD_EXACT_GZIP_TIMING=1 SPZ_AMD_LZ_TIMING=1 timeout -k 10 300 ./spz_amd/bin/host_bench 10000000 3 3 1 > $O/host_bench_s11.json 2> $O/host_bench_s11.err; echo "host_bench rc=$?"; cat $O/host_bench_s11.json; grep -E "lz77\]|saveSpz\]|exactgz\]" $O/host_bench_s11.err | tail -n 20
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-whole-file > $O/bench10_$i.json 2> $O/bench10_$i.err || { echo "bench $i failed"; tail -n 5 $O/bench10_$i.err; exit 3; }
  python - $O/bench10_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
p = d["placement"]
print("probe", round(d["value"] / 1e9, 3), "G/s  dec", round(d["roofline"]["avg_launch_ms"], 4), round(d["roofline"]["frac"], 3), " enc", round(d["roofline_encode"]["avg_launch_ms"], 4), round(d["roofline_encode"]["frac"], 3), "| dec probe", p["decode_buffers"]["sh_placements_timed"], p["decode_buffers"]["probe_ms_chosen"], p["decode_buffers"]["probe_ms_slowest"], "| enc probe", p["encode_buffers"]["sh_placements_timed"], p["encode_buffers"]["probe_ms_chosen"], p["encode_buffers"]["probe_ms_slowest"])
PY
done
