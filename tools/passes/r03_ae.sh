#!/bin/bash
# small-input path, even geometry: work-item length again, and the per-launch timeline at 2^14
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
for ln in 16 15 14; do
  echo "== 2^$ln"
  python tools/ab_knobs.py --log-n $ln --reps 6 --iters 20 "MSM377_NARROW_SEG=8" "MSM377_NARROW_SEG=12" "MSM377_NARROW_SEG=16" "MSM377_NARROW_SEG=24" 2>&1 | grep -v amdgpu.ids || exit 1
done > $out/r03_sweep_narrow_seg3.txt 2>&1; cat $out/r03_sweep_narrow_seg3.txt
for ln in 14 16; do echo "== 2^$ln"; PLAIN=1 LOG_N=$ln bash tools/trace_one_msm.sh r03_trace_small2_$ln 2>&1 | tail -22 | cut -c1-110; done > $out/r03_trace_small2.txt 2>&1; cat $out/r03_trace_small2.txt
