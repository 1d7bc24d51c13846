#!/bin/bash
# longer randomized soaks at the head, other seeds
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
SOAK_SEED=777 SOAK_SECONDS=420 timeout -k 10 600 python tools/soak_g1.py > $out/r03_soak_long.txt 2>&1; rc=$?; tail -3 $out/r03_soak_long.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python tools/soak_ed.py > $out/r03_soak_ed.txt 2>&1; rc=$?; tail -3 $out/r03_soak_ed.txt; exit $rc
