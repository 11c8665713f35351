#!/bin/bash
# round 3: placement tests, bench lines from fresh processes with and without the placement probe, inflate coverage
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_placement.py tests/test_gpu_packed_device.py -x -q > $O/pytest_s7.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -n 8 $O/pytest_s7.log
[ $rc -eq 0 ] || exit $rc
for i in 1 2 3 4 5; do
  SPZ_AMD_LZ_TIMING=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-whole-file > $O/bench_probe_$i.json 2> $O/bench_probe_$i.err || { echo "bench probe $i failed"; tail -n 5 $O/bench_probe_$i.err; exit 3; }
  python - $O/bench_probe_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("probe", round(d["value"] / 1e9, 3), "G/s  dec", round(d["roofline"]["avg_launch_ms"], 4), round(d["roofline"]["frac"], 3), " enc", round(d["roofline_encode"]["avg_launch_ms"], 4), round(d["roofline_encode"]["frac"], 3), d["placement"].get("decode_buffers"), d["placement"].get("encode_buffers"))
PY
done
for i in 1 2 3; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-whole-file --placement plain > $O/bench_plain_$i.json 2> $O/bench_plain_$i.err || { echo "bench plain $i failed"; exit 4; }
  python - $O/bench_plain_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("plain", round(d["value"] / 1e9, 3), "G/s  dec", round(d["roofline"]["avg_launch_ms"], 4), round(d["roofline"]["frac"], 3), " enc", round(d["roofline_encode"]["avg_launch_ms"], 4), round(d["roofline_encode"]["frac"], 3))
PY
done
timeout -k 10 500 python tools/inflate_coverage.py --per-cell 2 > $O/inflate_coverage.json 2> $O/inflate_coverage.err; echo "coverage rc=$?"; python - <<'PY'
import json, os
d = json.load(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "inflate_coverage.json")))
print(d["members"], d["by_writer"]); print(d["by_texture"])
PY
