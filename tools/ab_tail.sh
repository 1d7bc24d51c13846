#!/bin/bash
# Host-tail A/B on one box: thread counts and worker polling at 2^20 and 2^12, then the small-input sweep.
set -e
python tools/ab_knobs.py --log-n 20 --reps 6 --iters 10 "MSM377_TAIL_THREADS=1" "MSM377_TAIL_THREADS=6 MSM377_TAIL_SPIN_US=0" "MSM377_TAIL_THREADS=6" "MSM377_TAIL_THREADS=4" "MSM377_TAIL_THREADS=8" > gpurun_out/ab_tail.txt 2>&1
python tools/ab_knobs.py --log-n 12 --reps 6 --iters 20 "MSM377_TAIL_THREADS=1" "MSM377_TAIL_THREADS=6 MSM377_TAIL_SPIN_US=0" "MSM377_TAIL_THREADS=6" "MSM377_TAIL_THREADS=4" "MSM377_TAIL_THREADS=8" >> gpurun_out/ab_tail.txt 2>&1
python tools/sweep_small.py > gpurun_out/sweep_small.txt 2>&1
