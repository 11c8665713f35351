#!/bin/bash
# round 2, GPU session B: full parity suite, interleave policy A/B, host-path probe + bench, per-kernel profiles.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -q -x > $O/pytest_gpu_r02b.log 2>&1
rc=$?
tail -n 5 $O/pytest_gpu_r02b.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
timeout -k 10 120 ./build/pcie_probe > $O/pcie_probe_r02b.jsonl 2>&1 || { echo "pcie_probe failed"; tail -n 5 $O/pcie_probe_r02b.jsonl; exit 3; }
cat $O/pcie_probe_r02b.jsonl
timeout -k 10 200 ./spz_amd/bin/host_bench 10000000 3 4 > $O/host_bench_r02b.json 2>&1 || { echo "host_bench failed"; tail -n 5 $O/host_bench_r02b.json; exit 3; }
cat $O/host_bench_r02b.json
V=quat_ieee,quat_fast,il_enc,il_dec,il_both,policy,u8_policy,u8_il_dec,policy_b
timeout -k 10 200 python tools/tune.py run --deg 3 --variants $V > $O/tune_r02b_sh3.jsonl 2>&1 || { echo "tune sh3 failed"; tail -n 5 $O/tune_r02b_sh3.jsonl; exit 3; }
timeout -k 10 200 python tools/tune.py run --deg 2 --variants $V > $O/tune_r02b_sh2.jsonl 2>&1 || { echo "tune sh2 failed"; tail -n 5 $O/tune_r02b_sh2.jsonl; exit 3; }
timeout -k 10 200 python tools/tune.py run --deg 0 --variants $V > $O/tune_r02b_sh0.jsonl 2>&1 || { echo "tune sh0 failed"; tail -n 5 $O/tune_r02b_sh0.jsonl; exit 3; }
timeout -k 10 200 python tools/tune.py run --deg 0 --version 2 --points 1000000 --rounds 30 --batch 20 --variants $V > $O/tune_r02b_cfg2.jsonl 2>&1 || { echo "tune cfg2 failed"; tail -n 5 $O/tune_r02b_cfg2.jsonl; exit 3; }
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out","tune_r02b_*.jsonl"))):
    print(os.path.basename(f))
    for l in open(f):
        if l.startswith("{"):
            r=json.loads(l); print(f"  {r['variant']:12s} enc {r['enc_ms_med']:.4f} ({r['enc_frac_of_8TBps']:.3f})  dec {r['dec_ms_med']:.4f} ({r['dec_frac_of_8TBps']:.3f})  cold {r['dec_cold_ms_med']:.4f}")
PY
bash tools/gpu_profile_all.sh r02b
