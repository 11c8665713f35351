#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
TAG=${1:-r02e}
mkdir -p $O
cd $R
for cfg in "4 0" "2 0" "8 0" "4 1" "8 1" "16 1"; do
  set -- $cfg
  SPZ_AMD_PREFAULT_THREADS=$1 SPZ_AMD_PREFAULT_JOIN_FIRST=$2 timeout -k 10 200 ./spz_amd/bin/host_bench 10000000 3 3 > $O/host_bench_${TAG}_t$1_j$2.json 2>&1 || { echo "host_bench failed"; exit 3; }
  echo "threads=$1 join_first=$2: $(python3 -c "import json,sys; d=json.load(open('$O/host_bench_${TAG}_t$1_j$2.json')); print('pack fresh', d['pack_to_stream_fresh_vector_s'], 'unpack', d['unpack_from_stream_s'])")"
done
V=quat_fast,il_enc,il_dec,il_both,policy,policy_b
timeout -k 10 200 python tools/tune.py run --deg 3 --variants $V > $O/tune_${TAG}_sh3.jsonl 2>&1 || { echo "tune sh3 failed"; tail -n 5 $O/tune_${TAG}_sh3.jsonl; exit 3; }
timeout -k 10 200 python tools/tune.py run --deg 0 --variants $V > $O/tune_${TAG}_sh0.jsonl 2>&1 || { echo "tune sh0 failed"; tail -n 5 $O/tune_${TAG}_sh0.jsonl; exit 3; }
timeout -k 10 200 python tools/tune.py run --deg 1 --variants $V > $O/tune_${TAG}_sh1.jsonl 2>&1 || { echo "tune sh1 failed"; tail -n 5 $O/tune_${TAG}_sh1.jsonl; exit 3; }
TAGX=$TAG python - <<'PY'
import json,glob,os
tag=os.environ.get("TAGX","r02e")
for f in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out",f"tune_{tag}_*.jsonl"))):
    print(os.path.basename(f))
    for l in open(f):
        if l.startswith("{"):
            r=json.loads(l); print(f"  {r['variant']:12s} enc {r['enc_ms_med']:.4f} ({r['enc_frac_of_8TBps']:.3f})  dec {r['dec_ms_med']:.4f} ({r['dec_frac_of_8TBps']:.3f})  cold {r['dec_cold_ms_med']:.4f}")
PY
bash tools/gpu_profile_all.sh $TAG || exit 4
timeout -k 10 400 python tools/gzip_scaling.py 10000000 8 16 32 64 > $O/gzip_scaling_$TAG.jsonl 2>&1; cat $O/gzip_scaling_$TAG.jsonl
