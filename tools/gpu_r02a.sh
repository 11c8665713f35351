#!/bin/bash
# round 2, GPU session A: proof obligations of the fast quaternion arithmetic, parity, then A/B timings.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fast_divisions or quaternion or v2_encode or large_random or whole_float or outside_the_defined or fuzz_sizes or baseline_config2" > $O/pytest_r02a.log 2>&1
rc=$?
tail -n 5 $O/pytest_r02a.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
V=quat_ieee,quat_fast,il_enc,il_both,il_ieee,u2,u8,u2_il,quat_fast_b
timeout -k 10 200 python tools/tune.py run --deg 0 --variants $V > $O/tune_r02a_sh0.jsonl 2>&1 || { echo "tune sh0 failed"; tail -n 5 $O/tune_r02a_sh0.jsonl; exit 3; }
timeout -k 10 200 python tools/tune.py run --deg 3 --variants $V > $O/tune_r02a_sh3.jsonl 2>&1 || { echo "tune sh3 failed"; tail -n 5 $O/tune_r02a_sh3.jsonl; exit 3; }
timeout -k 10 200 python tools/tune.py run --deg 1 --variants $V > $O/tune_r02a_sh1.jsonl 2>&1 || { echo "tune sh1 failed"; tail -n 5 $O/tune_r02a_sh1.jsonl; exit 3; }
timeout -k 10 200 python tools/tune.py run --deg 0 --version 2 --points 1000000 --rounds 40 --variants $V,u1 > $O/tune_r02a_cfg2.jsonl 2>&1 || { echo "tune cfg2 failed"; tail -n 5 $O/tune_r02a_cfg2.jsonl; exit 3; }
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out","tune_r02a_*.jsonl"))):
    print(os.path.basename(f))
    for l in open(f):
        if l.startswith("{"):
            r=json.loads(l); print(f"  {r['variant']:12s} enc {r['enc_ms_med']:.4f} ({r['enc_frac_of_8TBps']:.3f})  dec {r['dec_ms_med']:.4f} ({r['dec_frac_of_8TBps']:.3f})  cold {r['dec_cold_ms_med']:.4f}")
PY
