#!/bin/bash
mkdir -p gpurun_out/dp
python bench.py --no-cpu-baseline --steps 300 --warmup 30 > gpurun_out/dp/plain.json 2> gpurun_out/dp/plain.err
bash scripts/dp1.sh --steps 300 --warmup 30 > gpurun_out/dp/dp_yelp.json 2> gpurun_out/dp/dp_yelp.err
bash scripts/dp1.sh --workload synth-10M --dim 256 --steps 12 --warmup 4 > gpurun_out/dp/dp_c5.json 2> gpurun_out/dp/dp_c5.err
tail -c 500 gpurun_out/dp/dp_c5.json; tail -n 5 gpurun_out/dp/dp_c5.err
