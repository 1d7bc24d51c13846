#!/bin/bash
# Round 3, final profile pass at the head (even windows, twin batches, wide sort v2, no event records): profiles/r03_final/
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
for rep in 1 2 3; do
  (cd $root/ab/r02_tree && python bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('r02 head  %.4f ms  acc %.4f  %s' % (d['value'], d['roofline']['kernel_ms'], {k: round(v,3) for k,v in d['stages_ms'].items()}))")
  (cd $root && python bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('r03 head  %.4f ms  acc %.4f  %s' % (d['value'], d['roofline']['kernel_ms'], {k: round(v,3) for k,v in d['stages_ms'].items()}))")
done > $out/r03_ab_bench.txt 2>&1; cat $out/r03_ab_bench.txt
bash tools/profile_round.sh r03_final > $out/r03_profile_round.log 2>&1; tail -15 $out/r03_profile_round.log
bash tools/trace_one_msm.sh r03_trace_final LOG_N=20 PLAIN=1 > $out/r03_trace_final.txt 2>&1; cat $out/r03_trace_final.txt
python tools/sweep_small.py > $out/r03_sweep_small.txt 2>&1; grep -v amdgpu $out/r03_sweep_small.txt
python tools/stage_fixed.py 20 10 2>&1 | grep -v amdgpu > $out/r03_stage_fixed_final.txt; cat $out/r03_stage_fixed_final.txt
for rep in 1 2; do
  (cd $root && python tools/h2d_one.py 20 15) 2>&1 | grep -v amdgpu
done > $out/r03_upload_final.txt 2>&1; cat $out/r03_upload_final.txt
python tools/time_shard.py --log-n 20 > $out/r03_time_shard_2e20.txt 2>&1; grep -v amdgpu $out/r03_time_shard_2e20.txt | tail -25
