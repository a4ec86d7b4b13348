#!/bin/bash
mkdir -p gpurun_out/r2l
for sp in 128 96 64 48; do
  timeout 300 python bench.py --no-cpu-baseline --hbm-leg on --epoch-leg off --split $sp > gpurun_out/r2l/bench_split$sp.json 2> gpurun_out/r2l/bench_split$sp.err
  timeout 300 python bench.py --no-cpu-baseline --workload amazon-book --hbm-leg off --epoch-leg off --split $sp > gpurun_out/r2l/bench_amazon_split$sp.json 2> gpurun_out/r2l/bench_amazon_split$sp.err
done
for sp in 128 96 64 48; do for f in bench_split$sp bench_amazon_split$sp; do echo "== $f"; python scripts/brief.py < gpurun_out/r2l/$f.json; done; done
