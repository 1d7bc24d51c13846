#!/bin/bash
# narrow-window path: work-item length at 2^15 and 2^16 (interleaved A/B in one process)
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
for ln in 16 15 14 13; do
  echo "== 2^$ln"
  python tools/ab_knobs.py --log-n $ln --reps 6 --iters 20 "MSM377_NARROW_SEG=8" "MSM377_NARROW_SEG=16" "MSM377_NARROW_SEG=32" "MSM377_NARROW_SEG=64" 2>&1 | grep -v amdgpu.ids || exit 1
done > $out/r03_sweep_narrow_seg2.txt 2>&1; cat $out/r03_sweep_narrow_seg2.txt
