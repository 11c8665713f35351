#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
TAG=${1:-r02n}
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu_$TAG.log 2>&1
rc=$?
tail -n 6 $O/pytest_gpu_$TAG.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
V=seq,policy,enc_nts,dec_ld,dec_st,policy_b
for deg in 3 0; do
timeout -k 10 200 python tools/tune.py run --deg $deg --rounds 11 --variants $V > $O/tune_${TAG}_sh$deg.jsonl 2>&1 || { echo "tune sh$deg failed"; tail -n 5 $O/tune_${TAG}_sh$deg.jsonl; exit 3; }
done
TAGX=$TAG python - <<'PY'
import json,glob,os
tag=os.environ.get("TAGX")
for f in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out",f"tune_{tag}_*.jsonl"))):
    print(os.path.basename(f))
    for l in open(f):
        if l.startswith("{"):
            r=json.loads(l); print(f"  {r['variant']:12s} enc {r['enc_ms_med']:.4f} ({r['enc_frac_of_8TBps']:.3f})  dec {r['dec_ms_med']:.4f} ({r['dec_frac_of_8TBps']:.3f})  cold {r['dec_cold_ms_med']:.4f}  pair {r['pair_ms_med']:.4f}")
PY
