#!/bin/bash
# One gpurun call -> everything profiles/<tag>/ needs (run from the repo root ON THE GPU BOX):
#   bench.json               python bench.py   (default steps / warm-up, free-running clocks, with the CPU baseline)
#   bench_ed / bench_fixed64 / bench_fixed64_wide / bench_driver_flags / bench_r01_protocol .json   the side workloads and protocols
#   kernel_stats.csv         rocprofv3 --kernel-trace --stats of `bench.py --steps 20 --warmup 5 --no-cpu-baseline`
#   bench_under_rocprof.log  the bench line printed inside that profiled run (its roofline.kernel_ms must agree)
#   pmc_summary.json         three separate --pmc passes (FETCH_SIZE; WRITE_SIZE; SQ_* + GRBM), tools/summarize_pmc.py
# usage: tools/profile_round.sh <tag>
set -e
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
python3 bench.py > $out/bench.json 2> $out/bench.err
python3 bench.py --workload ed > $out/bench_ed.json 2>> $out/bench.err
python3 bench.py --workload fixed64 --steps 20 --warmup 2 > $out/bench_fixed64.json 2>> $out/bench.err
MSM377_BENCH_PRECOMPUTE=20 python3 bench.py --workload fixed64 --steps 20 --warmup 2 > $out/bench_fixed64_wide.json 2>> $out/bench.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_driver_flags.json 2>> $out/bench.err
python3 bench.py --steps 10 --warmup 2 --setup-msms 0 --no-cpu-baseline > $out/bench_r01_protocol.json 2>> $out/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_under_rocprof.log 2> $out/rocprof_stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/rocprof_pmc1.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/rocprof_pmc2.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $out/pmc_sq -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/rocprof_pmc3.err
cd $root
python3 tools/summarize_pmc.py $out/pmc_summary.json $out/pmc_fetch $out/pmc_write $out/pmc_sq
f=$(find $out/stats -name "*kernel_stats.csv" | head -1)
cp "$f" $out/kernel_stats.csv
rm -rf $out/stats $out/pmc_fetch $out/pmc_write $out/pmc_sq
head -12 $out/kernel_stats.csv
cat $out/bench.json | cut -c1-400
