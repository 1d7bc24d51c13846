#!/bin/bash
# where does the batched affine conversion start to pay?  (MSM377_AFFINE_MIN, default 2^20)
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
for ln in 19 18; do
  echo "== 2^$ln"
  python tools/ab_knobs.py --log-n $ln --reps 6 --iters 10 "MSM377_AFFINE_MIN=1048576" "MSM377_AFFINE_MIN=131072" 2>&1 | grep -v amdgpu.ids || exit 1
done > $out/r03_sweep_affine_min.txt 2>&1; cat $out/r03_sweep_affine_min.txt
