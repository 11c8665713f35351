#!/bin/bash
# round 2, final-state session: parity, host path, SH2/SH3 interleave A/B on this box, per-kernel rows, bench passes + bench line.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
TAG=${1:-r02}
mkdir -p $O
cd $R
timeout -k 10 700 python -m pytest tests -m gpu -q -x > $O/pytest_gpu_$TAG.log 2>&1
rc=$?
tail -n 6 $O/pytest_gpu_$TAG.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
timeout -k 10 200 ./spz_amd/bin/host_bench 10000000 3 4 1 > $O/host_bench_$TAG.json 2>&1 || { echo "host_bench failed"; tail -n 5 $O/host_bench_$TAG.json; exit 3; }
cat $O/host_bench_$TAG.json
V=seq,il_g1,il_g8,policy,policy_b
for deg in 3 2; do
timeout -k 10 200 python tools/tune.py run --deg $deg --rounds 11 --variants $V > $O/tune_${TAG}_sh$deg.jsonl 2>&1 || { echo "tune sh$deg failed"; tail -n 5 $O/tune_${TAG}_sh$deg.jsonl; exit 3; }
done
timeout -k 10 200 python tools/tune.py run --deg 0 --version 2 --points 1000000 --rounds 30 --batch 20 --variants quat_ieee,seq,policy,policy_b > $O/tune_${TAG}_cfg2.jsonl 2>&1 || { echo "tune cfg2 failed"; exit 3; }
TAGX=$TAG python - <<'PY'
import json,glob,os
tag=os.environ.get("TAGX")
for f in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out",f"tune_{tag}_*.jsonl"))):
    print(os.path.basename(f))
    for l in open(f):
        if l.startswith("{"):
            r=json.loads(l); print(f"  {r['variant']:12s} enc {r['enc_ms_med']:.4f} ({r['enc_frac_of_8TBps']:.3f})  dec {r['dec_ms_med']:.4f} ({r['dec_frac_of_8TBps']:.3f})  cold {r['dec_cold_ms_med']:.4f}")
PY
bash tools/gpu_profile_all.sh $TAG || exit 4
SKIP_TUNE=1 SKIP_PYTEST=1 bash tools/gpu_round.sh $TAG
