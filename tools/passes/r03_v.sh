#!/bin/bash
# no event records between the kernels, no k_work_scan launch: full GPU suite, then A/B against the previous head at 2^20 and small sizes
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
timeout -k 10 1000 python -m pytest tests -x -q -m gpu -k "not 2_22 and not config5" > $out/r03_pytest_v.txt 2>&1; rc=$?; tail -5 $out/r03_pytest_v.txt
[ $rc -eq 0 ] || exit $rc
bash tools/ab_libs.sh 3 ab/libmsm377_head.so webgpu-msm-bls12-377_amd/csrc/libmsm377.so > $out/r03_ab_bubbles.txt 2>&1 || exit 1
for lib in ab/libmsm377_head.so webgpu-msm-bls12-377_amd/csrc/libmsm377.so; do echo "== $lib"; MSM377_LIB=$lib python tools/sweep_small.py 2>&1 | grep -v amdgpu; done >> $out/r03_ab_bubbles.txt 2>&1
cat $out/r03_ab_bubbles.txt | cut -c1-200
