#!/bin/bash
# wide-window sort: parallel scan kernels, chunk counts; parity of the wide path
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
python -m pytest tests/test_g1_parity_gpu.py -x -q -k "precomputed or wide_windows or config5" > $out/r03_pytest_l.txt 2>&1; tail -3 $out/r03_pytest_l.txt
for ch in 256 512 1024; do
  echo "MSM377_WIDE_CHUNKS=$ch"; MSM377_WIDE_CHUNKS=$ch python tools/stage_fixed.py 20 10 2>&1 | grep "20-bit"
done > $out/r03_wide_chunks.txt 2>&1; cat $out/r03_wide_chunks.txt
