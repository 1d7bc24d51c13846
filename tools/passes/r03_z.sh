#!/bin/bash
# timeline of a 2^19 MSM with affine records (where does the conversion chain's latency sit?), plus the new batch tests
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
timeout -k 10 300 python -m pytest tests/test_g1_parity_gpu.py -x -q -k "batch_pipeline" > $out/r03_pytest_z.txt 2>&1; rc=$?; tail -3 $out/r03_pytest_z.txt
[ $rc -eq 0 ] || exit $rc
for ln in 19 18; do
echo "== 2^$ln affine"; PLAIN=1 LOG_N=$ln bash tools/trace_one_msm.sh r03_trace_aff_$ln MSM377_AFFINE_MIN=131072 2>&1 | tail -28 | cut -c1-110
done > $out/r03_trace_affine_small.txt 2>&1; cat $out/r03_trace_affine_small.txt
