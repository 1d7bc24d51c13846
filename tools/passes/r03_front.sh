#!/bin/bash
# Round 3, first GPU pass over the new front-end / reduction kernels: parity suite with the new defaults, interleaved A/B of
# every knob against the round-2 kernels, and the per-launch timeline.  Run from the repo root on the GPU box.
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
python -m pytest tests -m gpu -x -q > $out/r03_pytest_a.txt 2>&1; tail -3 $out/r03_pytest_a.txt
python tools/ab_knobs.py --log-n 20 --reps 6 --iters 8 \
  "MSM377_TREE_LDS=0 MSM377_SORT_STAGED=0 MSM377_AFF_PARTS=1" \
  "MSM377_TREE_LDS=1 MSM377_SORT_STAGED=0 MSM377_AFF_PARTS=1" \
  "MSM377_TREE_LDS=0 MSM377_SORT_STAGED=1 MSM377_AFF_PARTS=1" \
  "MSM377_TREE_LDS=0 MSM377_SORT_STAGED=2 MSM377_AFF_PARTS=1" \
  "MSM377_TREE_LDS=0 MSM377_SORT_STAGED=0 MSM377_AFF_PARTS=2" \
  "MSM377_TREE_LDS=1 MSM377_SORT_STAGED=3 MSM377_AFF_PARTS=2" > $out/r03_ab_front.txt 2>&1
cat $out/r03_ab_front.txt
bash tools/trace_one_msm.sh r03_trace_a LOG_N=20 PLAIN=1 > $out/r03_trace_a.txt 2>&1
cat $out/r03_trace_a.txt
