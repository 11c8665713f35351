#!/bin/bash
# decode time against buffer placement: bash tools/gpu_align_probe.sh [trials] [variants] [fresh]
cd ${GRAFT_REPO_ROOT:-.} && timeout -k 10 600 python tools/placement_probe.py ${1:-48} ${2:-seq,policy} ${3:-} > gpurun_out/placement_probe.jsonl 2>&1
python3 - <<'PY'
import json
rows=[json.loads(l) for l in open('gpurun_out/placement_probe.jsonl') if l.startswith('{')]
if rows:
    names=[k[:-7] for k in rows[0] if k.endswith('_dec_ms')]
    print('trial  ' + '  '.join(n.rjust(9) for n in names) + '   (decode ms)')
    for r in rows: print(str(r['trial']).rjust(5) + '  ' + '  '.join(f"{r[n+'_dec_ms']:9.4f}" for n in names))
else:
    print(open('gpurun_out/placement_probe.jsonl').read()[-1500:])
PY
