#!/bin/bash
# Round 3: the conversion in lazy field forms -- parity suite, A/B against the previous build (ab/libmsm377_prev.so), the
# conversion alone, timeline.
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
python -m pytest tests -m gpu -x -q > $out/r03_pytest_j.txt 2>&1; tail -3 $out/r03_pytest_j.txt
bash tools/ab_libs.sh 3 $root/ab/libmsm377_prev.so $root/webgpu-msm-bls12-377_amd/csrc/libmsm377.so 2>&1 | grep -v amdgpu > $out/r03_ab_lazyconv.txt
for lib in ab/libmsm377_prev.so webgpu-msm-bls12-377_amd/csrc/libmsm377.so; do
  echo "== $lib" >> $out/r03_ab_lazyconv.txt
  MSM377_LIB=$root/$lib bash tools/prof_set_bases.sh r03_prof_lazy_$(basename $(dirname $lib)) 2>&1 | grep -v amdgpu | cut -c1-60,180-330 >> $out/r03_ab_lazyconv.txt
done
cat $out/r03_ab_lazyconv.txt
bash tools/trace_one_msm.sh r03_trace_lazy LOG_N=20 PLAIN=1 > $out/r03_trace_lazy.txt 2>&1; tail -28 $out/r03_trace_lazy.txt | head -16
