#!/bin/bash
# several contexts on one GPU: do independent fixed-base MSMs overlap?
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
export MSM377_TWIN_BATCH=0
for k in 2 3 4; do for bits in 20 0; do timeout -k 10 200 python tools/twin_probe.py 20 12 $bits $k || exit 1; done; done > $out/r03_twin_probe.txt 2>&1; grep table $out/r03_twin_probe.txt
