#!/bin/bash
# kernel trace of the wide-window path
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/r03_wide_n -- python3 $root/tools/run_wide.py 20 8 > $out/r03_wide_n.log 2>&1 &&
python3 - <<PY
import csv, glob
f = glob.glob("$out/r03_wide_n/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:22]:
    print(r["Name"][:90], r["Calls"], r["AverageNs"])
PY
