#!/bin/bash
# Round 3, fourth GPU pass: the whole parity suite (chunk-aware local sort, sort-once upload, two-pass wide sort, point
# sharding), stage times per table kind, the bench line with the upload-inclusive figure, fixed64 over the wide table.
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
python -m pytest tests -m gpu -x -q > $out/r03_pytest_d.txt 2>&1; tail -4 $out/r03_pytest_d.txt
python tools/stage_fixed.py 20 10 > $out/r03_stage_fixed_d.txt 2>&1; cat $out/r03_stage_fixed_d.txt
python bench.py > $out/r03_bench_d.json 2> $out/r03_bench_d.err; python3 -c "
import json; d=json.load(open('$out/r03_bench_d.json')); print(d['value'], d['stages_ms'], 'h2d', d.get('ms_incl_h2d'), d.get('ms_incl_h2d_first_call'), d['roofline']['kernel_ms'])"
MSM377_BENCH_PRECOMPUTE=20 python bench.py --workload fixed64 --steps 20 --warmup 2 > $out/r03_fixed64_wide_d.json 2> $out/r03_fixed64_wide_d.err; cut -c1-60,380-800 $out/r03_fixed64_wide_d.json
python tools/h2d_ab.py > $out/r03_h2d_sweep.txt 2>&1; cat $out/r03_h2d_sweep.txt
