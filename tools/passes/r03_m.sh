#!/bin/bash
# wide windows: mixed widths (6 x 20 + 7 x 19) and the per-window two-level sort; parity, stage times, kernel trace
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
timeout -k 10 500 python -m pytest tests/test_g1_parity_gpu.py -x -q -k "precomputed or wide_windows or config5" > $out/r03_pytest_m.txt 2>&1; rc=$?; tail -5 $out/r03_pytest_m.txt
[ $rc -eq 0 ] || exit $rc
python tools/stage_fixed.py 20 10 > $out/r03_stage_fixed_m.txt 2>&1 && cat $out/r03_stage_fixed_m.txt &&
MSM377_BENCH_PRECOMPUTE=20 python bench.py --workload fixed64 --steps 2 --warmup 1 > $out/r03_bench_fixed64_wide_m.json 2> $out/r03_bench_fixed64_wide_m.err && cat $out/r03_bench_fixed64_wide_m.json &&
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/r03_wide_m -- python3 $root/tools/run_wide.py > $out/r03_wide_m.log 2>&1 &&
python3 - <<PY
import csv, glob
f = glob.glob("$out/r03_wide_m/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print(r["Name"][:70], r["Calls"], r["AverageNs"])
PY
