#!/bin/bash
# reduction tail out of LDS: parity (all curve policies, all geometries), then A/B at 2^20 and small sizes
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "not 2_22 and not config5 and not full_size" > $out/r03_pytest_ab.txt 2>&1; rc=$?; tail -3 $out/r03_pytest_ab.txt
[ $rc -eq 0 ] || exit $rc
MSM377_TAIL_FROM=6 timeout -k 10 300 python -m pytest tests/test_g1_parity_gpu.py tests/test_ed_parity_gpu.py -x -q -k "golden or ragged or 2_16 or reduction or precomputed" > $out/r03_pytest_ab6.txt 2>&1; rc=$?; tail -3 $out/r03_pytest_ab6.txt
[ $rc -eq 0 ] || exit $rc
for ln in 20 14 12; do
  echo "== 2^$ln"
  python tools/ab_knobs.py --log-n $ln --reps 6 --iters 12 "MSM377_TAIL_LDS=0" "MSM377_TAIL_LDS=1" "MSM377_TAIL_LDS=1 MSM377_TAIL_FROM=6" "MSM377_TAIL_LDS=1 MSM377_NARROW_TAIL_FROM=3" 2>&1 | grep -v amdgpu.ids || exit 1
done > $out/r03_ab_tail_lds.txt 2>&1; cat $out/r03_ab_tail_lds.txt
