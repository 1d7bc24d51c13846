#!/bin/bash
# even window geometry (13 x 16 + 3 x 15 bits): parity, then A/B of the headline bench and the host-buffer call
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
timeout -k 10 900 python -m pytest tests/test_g1_parity_gpu.py -x -q -k "not 2_22 and not config5" > $out/r03_pytest_q.txt 2>&1; rc=$?; tail -5 $out/r03_pytest_q.txt
[ $rc -eq 0 ] || exit $rc
for rep in 1 2 3; do for ev in 0 1; do
  echo -n "MSM377_EVEN_WINDOWS=$ev  "
  MSM377_EVEN_WINDOWS=$ev python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d.get('ms_incl_h2d'))" || exit 1
done; done > $out/r03_ab_even.txt 2>&1; cat $out/r03_ab_even.txt
