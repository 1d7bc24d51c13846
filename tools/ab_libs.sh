#!/bin/bash
# A/B of two (or more) builds of csrc/libmsm377.so on one box: bench.py per build, interleaved ROUNDS times.
#   tools/ab_libs.sh 3 ab/libmsm377_a.so webgpu-msm-bls12-377_amd/csrc/libmsm377.so
# Prints ms per MSM, the accumulation kernel and the stage times of every run.
rounds=$1; shift
for r in $(seq 1 "$rounds"); do
  for lib in "$@"; do
    MSM377_LIB=$lib python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('%-52s %.4f ms  acc %.4f  stages %s' % ('$lib', d['value'], d['roofline']['kernel_ms'], {k: round(v,3) for k,v in d['stages_ms'].items()}))"
  done
done
