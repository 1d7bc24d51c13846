#!/bin/bash
# affine conversion with 4 points per thread below 2^20: parity on the affected paths, then the threshold again
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
MSM377_AFFINE_MIN=1000 timeout -k 10 600 python -m pytest tests/test_g1_parity_gpu.py -x -q -k "ragged or golden or subgroup or exceptional or precomputed or fixed_base or 2_16 or even_window" > $out/r03_pytest_aa.txt 2>&1; rc=$?; tail -3 $out/r03_pytest_aa.txt
[ $rc -eq 0 ] || exit $rc
for ln in 19 18; do
  echo "== 2^$ln"
  python tools/ab_knobs.py --log-n $ln --reps 6 --iters 10 "MSM377_AFFINE_MIN=1048576" "MSM377_AFFINE_MIN=131072" 2>&1 | grep -v amdgpu.ids || exit 1
done > $out/r03_sweep_affine_min2.txt 2>&1; cat $out/r03_sweep_affine_min2.txt
