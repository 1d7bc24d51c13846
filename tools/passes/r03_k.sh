#!/bin/bash
# kernel stats of the wide-window path (single-pass sort) and a randomized parity soak at the head
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/r03_wide_stats1 -- python3 $root/tools/run_wide.py 20 6 > /dev/null 2>&1
cd $root
f=$(find $out/r03_wide_stats1 -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"].replace("msm377::","").replace("(anonymous namespace)::","").replace("void ","")
    if float(r["AverageNs"]) > 4000: print("%-58s calls %4s avg %9.1f us  min %8.1f max %8.1f" % (n[:58], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
timeout 400 python tools/soak_g1.py 2>&1 | grep -v amdgpu | tail -8 > $out/r03_soak_g1.txt; cat $out/r03_soak_g1.txt
