#!/bin/bash
# round 3, state of the round: the whole -m gpu suite, smoke, rocprofv3 kernel stats + PMC passes of bench.py, the bench
# line, ten more bench lines from fresh processes, host_bench with laps, rocprofv3 kernel stats of saveSpz + loadSpz
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
TAG=${1:-r03}
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu_$TAG.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -n 4 $O/pytest_gpu_$TAG.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1 || exit 2
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d $O/prof_stats_$TAG -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-whole-file > $O/bench_prof_$TAG.log 2>&1 || { echo "rocprof stats failed"; tail -n 5 $O/bench_prof_$TAG.log; exit 4; }
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof_fetch_$TAG -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-whole-file > $O/bench_fetch_$TAG.log 2>&1 || { echo "rocprof FETCH_SIZE failed"; tail -n 5 $O/bench_fetch_$TAG.log; exit 5; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/prof_write_$TAG -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-whole-file > $O/bench_write_$TAG.log 2>&1 || { echo "rocprof WRITE_SIZE failed"; tail -n 5 $O/bench_write_$TAG.log; exit 6; }
cd $R
timeout -k 10 500 python bench.py > $O/bench_$TAG.log 2> $O/bench_$TAG.err || { echo "bench failed"; tail -n 5 $O/bench_$TAG.err; exit 7; }
tail -n 1 $O/bench_$TAG.log | cut -c1-600
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-whole-file > $O/bench10_${TAG}_$i.json 2> $O/bench10_${TAG}_$i.err || { echo "bench $i failed"; exit 8; }
  python - $O/bench10_${TAG}_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
p = d["placement"]
print("probe", round(d["value"] / 1e9, 3), "G/s  dec", round(d["roofline"]["avg_launch_ms"], 4), round(d["roofline"]["frac"], 3), " enc", round(d["roofline_encode"]["avg_launch_ms"], 4), round(d["roofline_encode"]["frac"], 3), "| dec probe", p["decode_buffers"]["sh_placements_timed"], p["decode_buffers"]["probe_ms_chosen"], p["decode_buffers"]["probe_ms_slowest"])
PY
done
SPZ_AMD_LZ_TIMING=1 SPZ_AMD_EXACT_GZIP_TIMING=1 timeout -k 10 300 ./spz_amd/bin/host_bench 10000000 3 3 1 > $O/host_bench_$TAG.json 2> $O/host_bench_$TAG.err || { echo "host_bench failed"; exit 9; }
cat $O/host_bench_$TAG.json
timeout -k 10 300 ./spz_amd/bin/host_bench 10000000 3 3 1 > $O/host_bench_${TAG}_quiet.json 2>/dev/null; cat $O/host_bench_${TAG}_quiet.json
bash tools/gpu_container_stats.sh $TAG > $O/container_stats_run_$TAG.log 2>&1; tail -n 30 $O/container_stats_run_$TAG.log | cut -c1-200
