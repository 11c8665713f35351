#!/bin/bash
# One GPU-box session: parity tests, variant tuning, rocprofv3 kernel stats + PMC passes, bench.
# Run as:  gpurun --timeout 1100 -- 'bash tools/gpu_round.sh <tag>'
set -o pipefail
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
if [ "$SKIP_PYTEST" != "1" ]; then
  timeout -k 10 800 python -m pytest tests -m gpu -q > $O/pytest_gpu_$TAG.log 2>&1
  rc=$?
  echo "pytest rc=$rc" >> $O/pytest_gpu_$TAG.log
  tail -n 4 $O/pytest_gpu_$TAG.log
  if [ $rc -gt 1 ]; then echo "pytest was killed (rc=$rc): stopping"; exit $rc; fi
fi
if [ "$SKIP_TUNE" != "1" ]; then
  timeout -k 10 400 python tools/tune.py run ${TUNE_ARGS} > $O/tune_$TAG.log 2>&1 || { echo "tune failed"; tail -n 5 $O/tune_$TAG.log; exit 3; }
  cat $O/tune_$TAG.log
fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats_$TAG -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_prof_$TAG.log 2>&1 || { echo "rocprof stats failed"; tail -n 5 $O/bench_prof_$TAG.log; exit 4; }
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof_fetch_$TAG -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_fetch_$TAG.log 2>&1 || { echo "rocprof FETCH_SIZE failed"; tail -n 5 $O/bench_fetch_$TAG.log; exit 5; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/prof_write_$TAG -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_write_$TAG.log 2>&1 || { echo "rocprof WRITE_SIZE failed"; tail -n 5 $O/bench_write_$TAG.log; exit 6; }
# best effort: wave / instruction mix of the two kernels (not needed for the roofline numbers)
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/prof_sq_$TAG -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_sq_$TAG.log 2>&1 || echo "SQ pass failed (ignored)"
cd $R
timeout -k 10 400 python bench.py > $O/bench_$TAG.log 2>&1 || { echo "bench failed"; tail -n 5 $O/bench_$TAG.log; exit 7; }
tail -n 1 $O/bench_$TAG.log
find $O -name "*.csv" | head -20
