#!/bin/bash
# even geometry on the small-input path: parity, then A/B
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
timeout -k 10 800 python -m pytest tests/test_g1_parity_gpu.py tests/test_node_binding_gpu.py tests/test_bench_gpu.py -x -q -k "not 2_22 and not config5 and not full_size" > $out/r03_pytest_ad.txt 2>&1; rc=$?; tail -3 $out/r03_pytest_ad.txt
[ $rc -eq 0 ] || exit $rc
for ln in 16 15 14 13 12 10; do
  echo "== 2^$ln"
  python tools/ab_knobs.py --log-n $ln --reps 6 --iters 20 "MSM377_NARROW_EVEN=0" "MSM377_NARROW_EVEN=1" 2>&1 | grep -v amdgpu.ids || exit 1
done > $out/r03_ab_narrow_even.txt 2>&1; cat $out/r03_ab_narrow_even.txt
