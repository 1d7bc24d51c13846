#!/bin/bash
# quad-cooperative record loads (merge, quad tree level, global-memory tail) + narrow work items of 16: parity, A/B vs the head before
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "not 2_22 and not config5 and not full_size" > $out/r03_pytest_af.txt 2>&1; rc=$?; tail -3 $out/r03_pytest_af.txt
[ $rc -eq 0 ] || exit $rc
for lib in ab/libmsm377_head.so webgpu-msm-bls12-377_amd/csrc/libmsm377.so ab/libmsm377_head.so webgpu-msm-bls12-377_amd/csrc/libmsm377.so; do echo "== $lib"; MSM377_LIB=$lib python tools/sweep_small.py 2>&1 | grep -v amdgpu | cut -c1-60; done > $out/r03_ab_quad_loads.txt 2>&1
bash tools/ab_libs.sh 2 ab/libmsm377_head.so webgpu-msm-bls12-377_amd/csrc/libmsm377.so >> $out/r03_ab_quad_loads.txt 2>&1
cat $out/r03_ab_quad_loads.txt | cut -c1-200
