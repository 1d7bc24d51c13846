#!/bin/bash
# Round 3: A/B of the raised wave priority of the front-end kernels (s_setprio), timeline with it.
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
python -m pytest tests/test_g1_parity_gpu.py -x -q -k "golden or ragged or full_size_2_20" > $out/r03_pytest_i.txt 2>&1; tail -3 $out/r03_pytest_i.txt
python tools/ab_knobs.py --log-n 20 --reps 8 --iters 8 "MSM377_CONV_WAVE_PRIO=0 MSM377_FRONT_WAVE_PRIO=0" "MSM377_CONV_WAVE_PRIO=1 MSM377_FRONT_WAVE_PRIO=0" "MSM377_CONV_WAVE_PRIO=0 MSM377_FRONT_WAVE_PRIO=1" 2>&1 | grep -v amdgpu > $out/r03_ab_waveprio2.txt
python tools/ab_knobs.py --log-n 22 --reps 4 --iters 4 "MSM377_CONV_WAVE_PRIO=0 MSM377_FRONT_WAVE_PRIO=0" "MSM377_CONV_WAVE_PRIO=1 MSM377_FRONT_WAVE_PRIO=0" "MSM377_CONV_WAVE_PRIO=0 MSM377_FRONT_WAVE_PRIO=1" 2>&1 | grep -v amdgpu >> $out/r03_ab_waveprio2.txt
cat $out/r03_ab_waveprio2.txt
bash tools/trace_one_msm.sh r03_trace_convprio LOG_N=20 PLAIN=1 > $out/r03_trace_convprio.txt 2>&1; tail -28 $out/r03_trace_convprio.txt | head -16
