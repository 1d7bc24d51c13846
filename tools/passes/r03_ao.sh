#!/bin/bash
# host-buffer call with ONE continuous staged upload: parity, then A/B against the head before
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
timeout -k 10 600 python -m pytest tests/test_g1_parity_gpu.py tests/test_ed_parity_gpu.py tests/test_node_binding_gpu.py tests/test_bench_gpu.py -x -q -k "host_buffers or golden or compute_msm or node or ragged or bench or even_window" > $out/r03_pytest_ao.txt 2>&1; rc=$?; tail -3 $out/r03_pytest_ao.txt
[ $rc -eq 0 ] || exit $rc
for rep in 1 2 3; do
  for lib in ab/libmsm377_head.so webgpu-msm-bls12-377_amd/csrc/libmsm377.so; do echo -n "$lib  "; MSM377_LIB=$lib python tools/h2d_one.py 20 15 2>&1 | grep -v amdgpu; done
done > $out/r03_ab_upload_stream.txt 2>&1
for ch in 4 5 6; do for sp in 20 30; do echo -n "chunks $ch split $sp  "; MSM377_UPLOAD_CHUNKS=$ch MSM377_UPLOAD_SPLIT=$sp python tools/h2d_one.py 20 15 2>&1 | grep -v amdgpu; done; done >> $out/r03_ab_upload_stream.txt 2>&1
cat $out/r03_ab_upload_stream.txt
