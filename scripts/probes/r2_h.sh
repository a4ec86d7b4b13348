#!/bin/bash
# round 2 profiles: kernel traces (stats) and PMC passes of the end-of-session kernels; bench lines of every config
mkdir -p gpurun_out/r2h
python bench.py > gpurun_out/r2h/bench_default.json 2> gpurun_out/r2h/bench_default.err
bash scripts/prof.sh r02_yelp
bash scripts/prof.sh r02_synth1m --workload synth-1M --steps 40 --warmup 5
bash scripts/prof.sh r02_amazon --workload amazon-book
bash scripts/pmc.sh r02_yelp --separate-adam > gpurun_out/r2h/pmc_yelp.log 2>&1
python scripts/traffic_json.py gpurun_out/pmc_r02_yelp/summary.json yelp2018 64 > gpurun_out/r2h/traffic_yelp.log 2>&1
bash scripts/pmc.sh r02_synth1m --workload synth-1M --separate-adam > gpurun_out/r2h/pmc_synth1m.log 2>&1
python scripts/traffic_json.py gpurun_out/pmc_r02_synth1m/summary.json synth-1M 64 > gpurun_out/r2h/traffic_synth1m.log 2>&1
bash scripts/pmc.sh r02_amazon --workload amazon-book --separate-adam > gpurun_out/r2h/pmc_amazon.log 2>&1
python scripts/traffic_json.py gpurun_out/pmc_r02_amazon/summary.json amazon-book 64 > gpurun_out/r2h/traffic_amazon.log 2>&1
cp profiles/r02/traffic_*.json gpurun_out/r2h/ 2>/dev/null
python bench.py --no-cpu-baseline --workload amazon-book > gpurun_out/r2h/bench_amazon.json 2> gpurun_out/r2h/bench_amazon.err
python bench.py --workload amazon-book --model SimGCL --batch 2048 --steps 400 --warmup 40 --cpu-seconds 8 > gpurun_out/r2h/bench_simgcl_amazon.json 2> gpurun_out/r2h/bench_simgcl_amazon.err
python bench.py --no-cpu-baseline --model MFBPR --batch 2048 > gpurun_out/r2h/bench_mfbpr.json 2> gpurun_out/r2h/bench_mfbpr.err
python bench.py --no-cpu-baseline --workload synth-1M --steps 60 --warmup 10 > gpurun_out/r2h/bench_synth1m.json 2> gpurun_out/r2h/bench_synth1m.err
python bench.py --no-cpu-baseline --workload synth-10M --dim 256 --steps 12 --warmup 4 > gpurun_out/r2h/bench_c5_single_gpu.json 2> gpurun_out/r2h/bench_c5_single_gpu.err
python scripts/eval_bench.py > gpurun_out/r2h/eval_yelp.txt 2>&1
python scripts/eval_bench.py amazon-book > gpurun_out/r2h/eval_amazon.txt 2>&1
IDG_TOPK_FORM=0 python scripts/eval_bench.py > gpurun_out/r2h/eval_yelp_alternating_kernel.txt 2>&1
python scripts/rows_kernel_probe.py > gpurun_out/r2h/restricted_probe.txt 2>&1
grep -h "evaluation\|identical" gpurun_out/r2h/eval_*.txt; grep "us$" gpurun_out/r2h/restricted_probe.txt
for f in bench_default bench_amazon bench_simgcl_amazon bench_mfbpr bench_synth1m bench_c5_single_gpu; do echo "== $f"; python scripts/brief.py < gpurun_out/r2h/$f.json; tail -n 1 gpurun_out/r2h/$f.err; done
ls gpurun_out/prof_r02_yelp/*/ 2>/dev/null | head; cat gpurun_out/r2h/traffic_yelp.log | head -30
