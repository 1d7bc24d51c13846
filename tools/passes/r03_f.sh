#!/bin/bash
# Round 3, sixth GPU pass: parity suite at the head, same-box A/B of the round-2 head (ab/r02_tree) against this tree.
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
python -m pytest tests -m gpu -x -q > $out/r03_pytest_f.txt 2>&1; tail -3 $out/r03_pytest_f.txt
for rep in 1 2 3; do
  (cd $root/ab/r02_tree && python tools/h2d_one.py 20 15) 2>&1 | grep -v amdgpu
  (cd $root && python tools/h2d_one.py 20 15) 2>&1 | grep -v amdgpu
done > $out/r03_ab_r02_vs_r03.txt 2>&1; cat $out/r03_ab_r02_vs_r03.txt
