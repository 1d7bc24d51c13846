#!/bin/bash
# Round 3, final pass at the head: whole GPU suite, soak, then profiles/r03_final (tools/passes/r03_w.sh)
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
timeout -k 10 900 python -m pytest tests -q -m gpu > $out/pytest_gpu_r03_final.txt 2>&1; rc=$?; tail -4 $out/pytest_gpu_r03_final.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/soak_g1.py > $out/r03_soak_final.txt 2>&1; rc=$?; tail -3 $out/r03_soak_final.txt
[ $rc -eq 0 ] || exit $rc
bash tools/passes/r03_w.sh
