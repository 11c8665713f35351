#!/bin/bash
# round 3: size sweep of saveSpz / loadSpz over the container routes; inflate tests + coverage after the stored-first change
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_inflate_device.py tests/test_parallel_inflate.py -x -q > $O/pytest_s8.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -n 6 $O/pytest_s8.log
timeout -k 10 300 python tools/inflate_coverage.py --per-cell 2 > $O/inflate_coverage2.json 2> $O/inflate_coverage2.err; echo "coverage rc=$?"
python - <<'PY'
import json, os
d = json.load(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "inflate_coverage2.json")))
print(d["members"], d["by_writer"]); print(d["by_texture"])
PY
timeout -k 10 700 python tools/size_sweep.py > $O/size_sweep.json 2> $O/size_sweep.err; echo "sweep rc=$?"; tail -n 40 $O/size_sweep.err | cut -c1-330
