#!/bin/bash
# Round 3, third GPU pass: wide-window parity, stage times per table kind, fixed64 with / without the deeper gather prefetch,
# the batched-affine prototype, per-rank times of both multi-GPU partitionings at 2^20 and 2^22.
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
python -m pytest tests/test_g1_parity_gpu.py -x -q -k "precomputed or wide_windows or fixed_base" > $out/r03_pytest_wide.txt 2>&1; tail -3 $out/r03_pytest_wide.txt
python tools/stage_fixed.py 20 10 > $out/r03_stage_fixed.txt 2>&1; cat $out/r03_stage_fixed.txt
MSM377_TABLE_PREFETCH=0 python tools/stage_fixed.py 20 10 > $out/r03_stage_fixed_pf0.txt 2>&1; cat $out/r03_stage_fixed_pf0.txt
for pf in 0 1; do
  MSM377_TABLE_PREFETCH=$pf MSM377_BENCH_PRECOMPUTE=20 python bench.py --workload fixed64 --steps 5 --warmup 1 --no-cpu-baseline > $out/r03_fixed64_wide_pf$pf.json 2> $out/r03_fixed64_wide_pf$pf.err
  cut -c1-100,400-700 $out/r03_fixed64_wide_pf$pf.json
done
./webgpu-msm-bls12-377_amd/csrc/microbench_affine 22 > $out/r03_microbench_affine.txt 2>&1; cat $out/r03_microbench_affine.txt
python tools/time_shard.py --log-n 20 > $out/r03_time_shard_2e20.txt 2>&1; cat $out/r03_time_shard_2e20.txt
python tools/time_shard.py --log-n 22 --iters 5 > $out/r03_time_shard_2e22.txt 2>&1; cat $out/r03_time_shard_2e22.txt
