#!/bin/bash
# Samples sclk / socket power / junction temperature while a sustained batch of MSMs runs (is k_accumulate
# clock- or power-limited?).  Output: gpurun_out/smi_samples.txt
(python bench.py --workload fixed64 --no-cpu-baseline --steps 30 --warmup 1 > gpurun_out/smi_bench.json 2> gpurun_out/smi_bench.err) &
BP=$!
for i in $(seq 1 200); do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Power \(W\)|junction" | sed 's/.*: //' | tr '\n' ' ' >> gpurun_out/smi_samples.txt
  echo >> gpurun_out/smi_samples.txt
  sleep 0.2
  kill -0 $BP 2>/dev/null || break
done
wait $BP
