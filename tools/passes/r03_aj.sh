#!/bin/bash
# reduction levels on lane quads, now that a quad reads a record once: from which level on?
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
python tools/ab_knobs.py --log-n 20 --reps 6 --iters 10 "MSM377_COOP_THREADS=131072" "MSM377_COOP_THREADS=200000" "MSM377_COOP_THREADS=330000" "MSM377_COOP_THREADS=530000" "MSM377_COOP_THREADS=800000" "MSM377_COOP_THREADS=1100000" 2>&1 | grep -v amdgpu.ids > $out/r03_sweep_coop.txt || exit 1
python tools/ab_knobs.py --log-n 17 --reps 6 --iters 10 "MSM377_COOP_THREADS=131072" "MSM377_COOP_THREADS=330000" "MSM377_COOP_THREADS=530000" "MSM377_COOP_THREADS=1100000" 2>&1 | grep -v amdgpu.ids >> $out/r03_sweep_coop.txt || exit 1
cat $out/r03_sweep_coop.txt
