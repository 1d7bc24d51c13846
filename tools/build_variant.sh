#!/bin/bash
# Build-time A/B: tools/build_variant.sh NAME [-DFLAG=VALUE ...]  ->  ab/libmsm377_NAME.so  (select with MSM377_LIB=..., tools/ab_libs.sh)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/webgpu-msm-bls12-377_amd/csrc
out=$root/ab
mkdir -p $out/obj_$name
for tu in sequencer host_tail capi; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function "$@" -c -o $out/obj_$name/$tu.o $src/$tu.hip &
done
wait
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -o $out/libmsm377_$name.so $out/obj_$name/*.o
rm -rf $out/obj_$name
echo built $out/libmsm377_$name.so
