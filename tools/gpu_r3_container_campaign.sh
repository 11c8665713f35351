#!/bin/bash
# round 3, container stage at the HEAD: tests, host bench, kernel stats, then the randomized device-writer campaign
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
bash tools/gpu_r3_container_final.sh ${1:-r03h} || exit $?
timeout -k 10 600 python tools/gzip_campaign.py --writer device --inputs 1500 --seed 41 > $O/gzip_campaign_device_r03.json 2> $O/gzip_campaign_device_r03.err; echo "campaign rc=$?"; cat $O/gzip_campaign_device_r03.json
