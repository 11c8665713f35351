#!/bin/bash
# generic A/B session:  bash tools/gpu_ab.sh <tag> <variants> [degrees]   (+ config 2 with 20 launches per timed region)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
TAG=${1:-ab}
V=${2:-seq,policy,policy_b}
DEGS=${3:-"3 0"}
mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "seeded or ragged or quaternion or large_random or fuzz_sizes or shards or full_size" > $O/pytest_gpu_$TAG.log 2>&1
rc=$?
tail -n 3 $O/pytest_gpu_$TAG.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
for deg in $DEGS; do
timeout -k 10 200 python tools/tune.py run --deg $deg --rounds 11 --variants $V > $O/tune_${TAG}_sh$deg.jsonl 2>&1 || { echo "tune sh$deg failed"; tail -n 5 $O/tune_${TAG}_sh$deg.jsonl; exit 3; }
done
timeout -k 10 200 python tools/tune.py run --deg 0 --version 2 --points 1000000 --rounds 30 --batch 20 --variants $V > $O/tune_${TAG}_cfg2.jsonl 2>&1 || { echo "tune cfg2 failed"; exit 3; }
timeout -k 10 200 python tools/tune.py run --deg 0 --version 2 --points 1000000 --rounds 30 --batch 1 --variants $V > $O/tune_${TAG}_cfg2single.jsonl 2>&1 || { echo "tune cfg2 single failed"; exit 3; }
TAGX=$TAG python - <<'PY'
import json,glob,os
tag=os.environ.get("TAGX")
for f in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out",f"tune_{tag}_*.jsonl"))):
    print(os.path.basename(f))
    for l in open(f):
        if l.startswith("{"):
            r=json.loads(l); print(f"  {r['variant']:12s} enc {r['enc_ms_med']:.4f} ({r['enc_frac_of_8TBps']:.3f})  dec {r['dec_ms_med']:.4f} ({r['dec_frac_of_8TBps']:.3f})  cold {r['dec_cold_ms_med']:.4f}  pair {r['pair_ms_med']:.4f}")
PY
