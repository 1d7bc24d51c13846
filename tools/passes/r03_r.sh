#!/bin/bash
# even window geometry: interleaved A/B in one process, 2^20 and 2^22, with stage columns; kernel trace of one MSM
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
python tools/ab_knobs.py --log-n 20 --reps 8 --iters 10 "MSM377_EVEN_WINDOWS=0" "MSM377_EVEN_WINDOWS=1" > $out/r03_ab_even.txt 2>&1 || exit 1
python tools/ab_knobs.py --log-n 22 --reps 4 --iters 4 "MSM377_EVEN_WINDOWS=0" "MSM377_EVEN_WINDOWS=1" >> $out/r03_ab_even.txt 2>&1 || exit 1
python tools/ab_knobs.py --log-n 18 --reps 6 --iters 10 "MSM377_EVEN_WINDOWS=0" "MSM377_EVEN_WINDOWS=1" >> $out/r03_ab_even.txt 2>&1 || exit 1
cat $out/r03_ab_even.txt
PLAIN=1 LOG_N=20 bash tools/trace_one_msm.sh r03_trace_even > $out/r03_trace_even.txt 2>&1; tail -30 $out/r03_trace_even.txt | cut -c1-120
