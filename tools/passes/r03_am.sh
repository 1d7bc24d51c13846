#!/bin/bash
# claimed shares in the tail pool: threaded-tail parity (all thread counts), A/B against the head before, and a run with CPU hogs beside it
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
timeout -k 10 700 python -m pytest tests/test_g1_parity_gpu.py tests/test_ed_parity_gpu.py -x -q -k "alternative or golden or ragged or 2_16 or fixed_base or full_size" > $out/r03_pytest_am.txt 2>&1; rc=$?; tail -3 $out/r03_pytest_am.txt
[ $rc -eq 0 ] || exit $rc
bash tools/ab_libs.sh 3 ab/libmsm377_head.so webgpu-msm-bls12-377_amd/csrc/libmsm377.so > $out/r03_ab_shares.txt 2>&1 || exit 1
# the same A/B with every CPU of the box's share kept busy by spinning processes (helper threads lose their CPUs)
ncpu=$(nproc); pids=""
for i in $(seq 1 $ncpu); do (timeout 120 python3 -c "while True: pass") & pids="$pids $!"; done
echo "== with $ncpu CPU hogs" >> $out/r03_ab_shares.txt
bash tools/ab_libs.sh 2 ab/libmsm377_head.so webgpu-msm-bls12-377_amd/csrc/libmsm377.so >> $out/r03_ab_shares.txt 2>&1
for p in $pids; do kill $p 2>/dev/null; done; wait 2>/dev/null
cat $out/r03_ab_shares.txt | cut -c1-215
