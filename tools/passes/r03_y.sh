#!/bin/bash
# the whole GPU suite at the head, then the randomized soak over the new paths
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
timeout -k 10 1000 python -m pytest tests -q -m gpu > $out/pytest_gpu_r03_final.txt 2>&1; rc=$?; tail -5 $out/pytest_gpu_r03_final.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python tools/soak_g1.py > $out/r03_soak_final.txt 2>&1; rc=$?; tail -5 $out/r03_soak_final.txt; exit $rc
