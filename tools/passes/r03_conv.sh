#!/bin/bash
# Round 3, second GPU pass: parity of the wide-window (20-bit) precomputed path, build-time variants of the batched affine
# conversion (alone and inside the MSM), fixed-base batch of 64 with the plain and the wide table.
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
python -m pytest tests/test_g1_parity_gpu.py -x -q -k "precomputed or wide or config5 or fixed_base" > $out/r03_pytest_wide.txt 2>&1; tail -5 $out/r03_pytest_wide.txt
for lib in webgpu-msm-bls12-377_amd/csrc/libmsm377.so ab/libmsm377_k4.so ab/libmsm377_u2.so; do
  echo "== $lib" >> $out/r03_conv.txt
  MSM377_LIB=$root/$lib bash tools/prof_set_bases.sh r03_prof_$(basename $lib .so) >> $out/r03_conv.txt 2>&1
done
bash tools/ab_libs.sh 2 $root/webgpu-msm-bls12-377_amd/csrc/libmsm377.so $root/ab/libmsm377_k4.so $root/ab/libmsm377_u2.so >> $out/r03_conv.txt 2>&1
cat $out/r03_conv.txt
for pre in 0 20; do
  MSM377_BENCH_PRECOMPUTE=$pre python bench.py --workload fixed64 --steps 3 --warmup 1 > $out/r03_fixed64_pre$pre.json 2> $out/r03_fixed64_pre$pre.err
  cut -c1-700 $out/r03_fixed64_pre$pre.json; tail -3 $out/r03_fixed64_pre$pre.err
done
