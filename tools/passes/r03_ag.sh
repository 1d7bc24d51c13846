#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
for ln in 14 16; do echo "== 2^$ln"; PLAIN=1 LOG_N=$ln bash tools/trace_one_msm.sh r03_trace_small3_$ln 2>&1 | tail -19 | cut -c1-110; done > $out/r03_trace_small3.txt 2>&1; cat $out/r03_trace_small3.txt
