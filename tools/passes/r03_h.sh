#!/bin/bash
# Round 3: A/B of the stream priorities and of the pre-woken inversion threads (front end of the 2^20 path), node tests.
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
python -m pytest tests/test_node_binding_gpu.py tests/test_g1_parity_gpu.py -x -q -k "js or upload_in_chunks or point_sharding" > $out/r03_pytest_h.txt 2>&1; tail -3 $out/r03_pytest_h.txt
python tools/ab_knobs.py --log-n 20 --reps 6 --iters 8 \
  "MSM377_STREAM_PRIO=1 MSM377_AFF_PREWAKE_US=0" \
  "MSM377_STREAM_PRIO=1 MSM377_AFF_PREWAKE_US=600" \
  "MSM377_STREAM_PRIO=2 MSM377_AFF_PREWAKE_US=0" \
  "MSM377_STREAM_PRIO=2 MSM377_AFF_PREWAKE_US=600" \
  "MSM377_STREAM_PRIO=0 MSM377_AFF_PREWAKE_US=600" 2>&1 | grep -v amdgpu > $out/r03_ab_prio.txt
cat $out/r03_ab_prio.txt
bash tools/trace_one_msm.sh r03_trace_prio2 LOG_N=20 PLAIN=1 MSM377_STREAM_PRIO=2 > $out/r03_trace_prio2.txt 2>&1; tail -28 $out/r03_trace_prio2.txt | head -16
