#!/bin/bash
# two contexts on one GPU: do independent fixed-base MSMs overlap?
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
for bits in 20 0; do timeout -k 10 200 python tools/twin_probe.py 20 16 $bits || exit 1; done > $out/r03_twin_probe.txt 2>&1; cat $out/r03_twin_probe.txt
