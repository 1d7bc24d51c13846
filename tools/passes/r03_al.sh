#!/bin/bash
# thread-level reduction levels with two pair-additions per thread (k_tree_step_x2): parity, A/B
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
MSM377_TREE_X2=1 MSM377_TREE_X2_MIN=1 timeout -k 10 500 python -m pytest tests/test_g1_parity_gpu.py tests/test_ed_parity_gpu.py -x -q -k "golden or ragged or 2_16 or reduction or precomputed or skewed or sharding_on_one" > $out/r03_pytest_al.txt 2>&1; rc=$?; tail -3 $out/r03_pytest_al.txt
[ $rc -eq 0 ] || exit $rc
python tools/ab_knobs.py --log-n 20 --reps 6 --iters 10 "MSM377_TREE_X2=0" "MSM377_TREE_X2=1" "MSM377_TREE_X2=1 MSM377_TREE_X2_MIN=250000" "MSM377_TREE_X2=1 MSM377_TREE_X2_MIN=60000" 2>&1 | grep -v amdgpu.ids > $out/r03_ab_tree_x2.txt || exit 1
python tools/ab_knobs.py --log-n 17 --reps 6 --iters 10 "MSM377_TREE_X2=0" "MSM377_TREE_X2=1" 2>&1 | grep -v amdgpu.ids >> $out/r03_ab_tree_x2.txt || exit 1
cat $out/r03_ab_tree_x2.txt
