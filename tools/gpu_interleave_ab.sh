#!/bin/bash
# quick session: what kind of box is this, and the grid-order A/B on it (sequential vs interleaved, run lengths)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
TAG=${1:-r02f}
mkdir -p $O
cd $R
{ rocm-smi --showmemorypartition --showcomputepartition --showclocks --showpower --showmaxpower --showperflevel 2>&1 | grep -v "^=" | head -40; rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock|Memory Properties|Size:" | head -20; } > $O/box_info_$TAG.txt 2>&1
grep -E "partition|sclk|mclk|fclk|Power|Marketing" $O/box_info_$TAG.txt | head -20
V=seq,il_g1,il_g4,il_g8,il_g16,il_g64,policy
for deg in 3 2 1 0; do
timeout -k 10 200 python tools/tune.py run --deg $deg --rounds 11 --variants $V > $O/tune_${TAG}_sh$deg.jsonl 2>&1 || { echo "tune sh$deg failed"; tail -n 5 $O/tune_${TAG}_sh$deg.jsonl; exit 3; }
done
TAGX=$TAG python - <<'PY'
import json,glob,os
tag=os.environ.get("TAGX")
for f in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out",f"tune_{tag}_*.jsonl"))):
    print(os.path.basename(f))
    for l in open(f):
        if l.startswith("{"):
            r=json.loads(l); print(f"  {r['variant']:12s} enc {r['enc_ms_med']:.4f} ({r['enc_frac_of_8TBps']:.3f})  dec {r['dec_ms_med']:.4f} ({r['dec_frac_of_8TBps']:.3f})  cold {r['dec_cold_ms_med']:.4f}")
PY
