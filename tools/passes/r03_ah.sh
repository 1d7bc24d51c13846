#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
for ln in 16 15 14 13; do
  echo "== 2^$ln"
  python tools/ab_knobs.py --log-n $ln --reps 6 --iters 20 "MSM377_NARROW_SEG=8" "MSM377_NARROW_SEG=10" "MSM377_NARROW_SEG=12" "MSM377_NARROW_SEG=16" 2>&1 | grep -v amdgpu.ids || exit 1
done > $out/r03_sweep_narrow_seg4.txt 2>&1; cat $out/r03_sweep_narrow_seg4.txt
