#!/bin/bash
# Kernel durations of the affine conversion alone: tools/prof_set_bases.sh <tag>
out=$GRAFT_REPO_ROOT/gpurun_out/$1
python3 tools/time_set_bases.py 20 20
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/time_set_bases.py 20 20 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $out -name "*kernel_stats.csv" | head -1)
grep -E "k_affine|Name" "$f" | cut -d, -f1-7 | sed 's/(anonymous namespace):://g' | cut -c1-200
