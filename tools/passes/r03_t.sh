#!/bin/bash
# per-launch timelines at 2^14, 2^16, 2^17 (where does a small MSM spend its time?)
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
for ln in 14 16 17; do
  echo "== 2^$ln"; PLAIN=1 LOG_N=$ln bash tools/trace_one_msm.sh r03_trace_small_$ln 2>&1 | tail -32 | cut -c1-110
done > $out/r03_trace_small.txt 2>&1
cat $out/r03_trace_small.txt
