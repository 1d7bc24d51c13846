set -e
python tools/ab_knobs.py --log-n 20 --reps 6 --iters 10 "MSM377_TAIL_THREADS=1" "MSM377_TAIL_THREADS=6 MSM377_TAIL_SPIN_US=0" "MSM377_TAIL_THREADS=6" "MSM377_TAIL_THREADS=4" "MSM377_TAIL_THREADS=8" > gpurun_out/ab_tail5.txt 2>&1
python tools/ab_knobs.py --log-n 12 --reps 6 --iters 20 "MSM377_TAIL_THREADS=1" "MSM377_TAIL_THREADS=6 MSM377_TAIL_SPIN_US=0" "MSM377_TAIL_THREADS=6" "MSM377_TAIL_THREADS=4" "MSM377_TAIL_THREADS=8" >> gpurun_out/ab_tail5.txt 2>&1
for t in 1 6; do MSM377_TAIL_THREADS=$t python bench.py --workload ed --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null >> gpurun_out/ab_tail5.txt; done
python -m pytest tests -m gpu -x -q > gpurun_out/pt_tail5.txt 2>&1
