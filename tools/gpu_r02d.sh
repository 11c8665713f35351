#!/bin/bash
# round 2, GPU session D: parity, host path (prefault with 4 threads), interleave A/B on this box, per-kernel rows with
# SQ counters and the box's reference copy rate, gzip writer scaling on the host.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
TAG=${1:-r02d}
mkdir -p $O
cd $R
timeout -k 10 700 python -m pytest tests -m gpu -q -x > $O/pytest_gpu_$TAG.log 2>&1
rc=$?
tail -n 6 $O/pytest_gpu_$TAG.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python -c "import spz_amd.spz as s; print('effective cpus', s._effective_cpu_count())"
for i in 1 2; do timeout -k 10 200 ./spz_amd/bin/host_bench 10000000 3 4 > $O/host_bench_${TAG}_$i.json 2>&1 || { echo "host_bench failed"; tail -n 5 $O/host_bench_${TAG}_$i.json; exit 3; }; cat $O/host_bench_${TAG}_$i.json; done
SPZ_AMD_PREFAULT_THREADS=0 timeout -k 10 200 ./spz_amd/bin/host_bench 10000000 3 3 > $O/host_bench_${TAG}_noprefault.json 2>&1; cat $O/host_bench_${TAG}_noprefault.json
V=quat_fast,il_enc,il_dec,il_both,policy,policy_b
timeout -k 10 200 python tools/tune.py run --deg 3 --variants $V > $O/tune_${TAG}_sh3.jsonl 2>&1 || { echo "tune sh3 failed"; tail -n 5 $O/tune_${TAG}_sh3.jsonl; exit 3; }
timeout -k 10 200 python tools/tune.py run --deg 0 --variants $V > $O/tune_${TAG}_sh0.jsonl 2>&1 || { echo "tune sh0 failed"; tail -n 5 $O/tune_${TAG}_sh0.jsonl; exit 3; }
timeout -k 10 200 python tools/tune.py run --deg 1 --variants $V > $O/tune_${TAG}_sh1.jsonl 2>&1 || { echo "tune sh1 failed"; tail -n 5 $O/tune_${TAG}_sh1.jsonl; exit 3; }
python - <<'PY'
import json,glob,os
tag=os.environ.get("TAGX","r02d")
for f in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out",f"tune_{tag}_*.jsonl"))):
    print(os.path.basename(f))
    for l in open(f):
        if l.startswith("{"):
            r=json.loads(l); print(f"  {r['variant']:12s} enc {r['enc_ms_med']:.4f} ({r['enc_frac_of_8TBps']:.3f})  dec {r['dec_ms_med']:.4f} ({r['dec_frac_of_8TBps']:.3f})  cold {r['dec_cold_ms_med']:.4f}")
PY
bash tools/gpu_profile_all.sh $TAG || exit 4
timeout -k 10 400 python tools/gzip_scaling.py 10000000 8 16 32 64 > $O/gzip_scaling_$TAG.jsonl 2>&1; cat $O/gzip_scaling_$TAG.jsonl
