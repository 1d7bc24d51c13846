#!/bin/bash
# divstep inversion on the host (tail + conversion): parity, then A/B against the previous head
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
timeout -k 10 600 python -m pytest tests/test_g1_parity_gpu.py tests/test_ed_parity_gpu.py tests/test_node_binding_gpu.py -x -q -k "not 2_22 and not config5 and not full_size and not alternative" > $out/r03_pytest_ac.txt 2>&1; rc=$?; tail -3 $out/r03_pytest_ac.txt
[ $rc -eq 0 ] || exit $rc
bash tools/ab_libs.sh 3 ab/libmsm377_head.so webgpu-msm-bls12-377_amd/csrc/libmsm377.so > $out/r03_ab_inv.txt 2>&1 || exit 1
for lib in ab/libmsm377_head.so webgpu-msm-bls12-377_amd/csrc/libmsm377.so; do echo "== $lib"; MSM377_LIB=$lib python tools/sweep_small.py 2>&1 | grep -v amdgpu; done >> $out/r03_ab_inv.txt 2>&1
cat $out/r03_ab_inv.txt | cut -c1-230
