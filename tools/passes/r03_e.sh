#!/bin/bash
# Round 3, fifth GPU pass: trace of the upload-inclusive call, kernel stats of the wide-window path, small-n work-item lengths.
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
python tools/upload_trace.py 20 > $out/r03_upload_trace.txt 2>&1; grep -v amdgpu $out/r03_upload_trace.txt | tail -12
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/r03_wide_stats -- python3 $root/tools/run_wide.py 20 6 > /dev/null 2>&1
cd $root
f=$(find $out/r03_wide_stats -name "*kernel_stats.csv" | head -1)
cut -d, -f1-4 "$f" | sed 's/msm377:://g; s/(anonymous namespace):://g' | cut -c1-70,150-260 | head -40
for seg in 8 12 16 24; do
  echo "NARROW_SEG=$seg"; MSM377_NARROW_SEG=$seg python tools/sweep_small.py 2>&1 | grep -E "2.1[2456] "
done > $out/r03_sweep_seg.txt 2>&1; cat $out/r03_sweep_seg.txt
