#!/bin/bash
# final-tree check after the fused NGCF layer kernels: whole GPU suite + smoke + default bench, then the epoch profiles
bash scripts/gpu_all.sh
sed -i 's/prof_r04c/prof_r04d/g' scripts/probes/r4_t.sh
bash scripts/probes/r4_t.sh
python scripts/eval_bench.py yelp2018 2>&1 | grep "ms per full"
