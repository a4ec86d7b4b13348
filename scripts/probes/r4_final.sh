#!/bin/bash
# final-tree refresh of profiles/r04: suite, default bench (two forms), kernel stats of the default run, the other configs' lines,
# EGCF / NGCF epoch kernel stats
mkdir -p gpurun_out/final
bash scripts/gpu_all.sh
cp gpurun_out/all/bench.json gpurun_out/final/bench_default.json
python bench.py --steps 20 --warmup 5 > gpurun_out/final/bench_default_steps20_warmup5.json 2>/dev/null
bash scripts/prof.sh r04_yelp --scale-point off > /dev/null 2>&1
python bench.py --model MFBPR --batch 2048 > gpurun_out/final/bench_mfbpr.json 2>/dev/null
python bench.py --workload amazon-book > gpurun_out/final/bench_amazon.json 2>/dev/null
python bench.py --workload amazon-book --model SimGCL --batch 2048 > gpurun_out/final/bench_simgcl_amazon.json 2>/dev/null
sed -i 's/prof_r04c/prof_r04z/g' scripts/probes/r4_t.sh
bash scripts/probes/r4_t.sh
for m in SGL SimGCL XSimGCL LightGCN MFBPR; do python scripts/e2e_epoch.py $m 4 2>&1 | grep "Training time" | tail -1; done
python scripts/eval_bench.py yelp2018 2>&1 | grep "ms per full"
