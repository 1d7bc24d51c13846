#!/bin/bash
# batches on a twin context: parity of every batch test, A/B of the fixed64 bench
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
timeout -k 10 700 python -m pytest tests/test_g1_parity_gpu.py -x -q -k "batch or fixed_base or precomputed or wide_windows or config5 or fallback or exceptional" > $out/r03_pytest_p.txt 2>&1; rc=$?; tail -5 $out/r03_pytest_p.txt
[ $rc -eq 0 ] || exit $rc
for rep in 1 2; do for tw in 0 1; do for pc in 0 20; do
  echo "MSM377_TWIN_BATCH=$tw MSM377_BENCH_PRECOMPUTE=$pc"
  MSM377_TWIN_BATCH=$tw MSM377_BENCH_PRECOMPUTE=$pc python bench.py --workload fixed64 --steps 2 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['verified'][:40])" || exit 1
done; done; done > $out/r03_ab_twin.txt 2>&1; cat $out/r03_ab_twin.txt
