#!/bin/bash
# staged upload with claimed pieces (caller copies too): parity of the host-buffer tests, A/B against the head before, and with CPU hogs
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
timeout -k 10 500 python -m pytest tests/test_g1_parity_gpu.py tests/test_ed_parity_gpu.py tests/test_node_binding_gpu.py -x -q -k "host_buffers or golden or compute_msm or node or ragged" > $out/r03_pytest_an.txt 2>&1; rc=$?; tail -3 $out/r03_pytest_an.txt
[ $rc -eq 0 ] || exit $rc
for rep in 1 2 3; do
  for lib in ab/libmsm377_head.so webgpu-msm-bls12-377_amd/csrc/libmsm377.so; do echo -n "$lib  "; MSM377_LIB=$lib python tools/h2d_one.py 20 15 2>&1 | grep -v amdgpu; done
done > $out/r03_ab_upload_claim.txt 2>&1
ncpu=$(nproc); pids=""
for i in $(seq 1 $ncpu); do (timeout 60 python3 -c "while True: pass") & pids="$pids $!"; done
echo "== with $ncpu CPU hogs" >> $out/r03_ab_upload_claim.txt
for lib in ab/libmsm377_head.so webgpu-msm-bls12-377_amd/csrc/libmsm377.so; do echo -n "$lib  "; MSM377_LIB=$lib timeout 100 python tools/h2d_one.py 20 15 2>&1 | grep -v amdgpu; done >> $out/r03_ab_upload_claim.txt 2>&1
for p in $pids; do kill $p 2>/dev/null; done; wait 2>/dev/null
cat $out/r03_ab_upload_claim.txt
