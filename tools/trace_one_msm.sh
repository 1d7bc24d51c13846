#!/bin/bash
# Per-launch timeline of the LAST MSM of a short bench run under rocprofv3 --kernel-trace (durations under the profiler's
# clock).  usage: tools/trace_one_msm.sh <tag> [LOG_N=12] [PLAIN=1] [ENV=VAL ...]
# PLAIN=1: tools/run_plain.py instead of bench.py -- the engine's stage timing stays off (its events put ~10 us gaps between the stages)
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
env "$@" true
for kv in "$@"; do export "$kv"; done
if [ "${PLAIN:-0}" = 1 ]; then
  rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/run_plain.py ${LOG_N:-20} > /dev/null 2>&1
else
  rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --log-n ${LOG_N:-20} --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
fi
cd $GRAFT_REPO_ROOT
f=$(find $out -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "k_decompose" in r["Kernel_Name"]][-1]
t0=int(rows[idx]["Start_Timestamp"])
for r in rows[max(0,idx-4):]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    name=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","")
    print("%8.1f .. %8.1f us  dur %7.1f  %s" % ((s-t0)/1000,(e-t0)/1000,(e-s)/1000,name[:70]))
PY
