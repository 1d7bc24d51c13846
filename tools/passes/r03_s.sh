#!/bin/bash
# even windows on the Edwards-BLS12 path: parity, A/B
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
timeout -k 10 900 python -m pytest tests/test_ed_parity_gpu.py tests/test_node_binding_gpu.py -x -q > $out/r03_pytest_s.txt 2>&1; rc=$?; tail -5 $out/r03_pytest_s.txt
[ $rc -eq 0 ] || exit $rc
for rep in 1 2 3; do for ev in 0 1; do
  echo -n "MSM377_EVEN_WINDOWS=$ev  "
  MSM377_EVEN_WINDOWS=$ev python bench.py --workload ed --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'])" || exit 1
done; done > $out/r03_ab_even_ed.txt 2>&1; cat $out/r03_ab_even_ed.txt
