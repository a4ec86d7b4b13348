#!/bin/bash
mkdir -p gpurun_out/dp
for nt in 0 1 2 4 3 7; do
IDG_NT=$nt python bench.py --no-cpu-baseline --steps 300 --warmup 30 > gpurun_out/dp/nt_$nt.json 2> gpurun_out/dp/nt_$nt.err
done
for nt in 0 1 3; do
IDG_NT=$nt python bench.py --no-cpu-baseline --workload synth-1M --steps 40 --warmup 5 > gpurun_out/dp/nt1m_$nt.json 2> gpurun_out/dp/nt1m_$nt.err
done
