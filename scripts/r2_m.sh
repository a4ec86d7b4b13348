#!/bin/bash
mkdir -p gpurun_out/r2m
IDG_TOPK_FORM=wave timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "topk or score" > gpurun_out/r2m/topk_wave.txt 2>&1; echo "rc=$?" >> gpurun_out/r2m/topk_wave.txt
for form in wave slab; do
  IDG_TOPK_FORM=$form timeout 300 python scripts/eval_bench.py > gpurun_out/r2m/eval_$form.txt 2>&1
  IDG_TOPK_FORM=$form timeout 300 python scripts/eval_bench.py amazon-book > gpurun_out/r2m/eval_amazon_$form.txt 2>&1
done
for nc in 2 3 4 6; do IDG_TOPK_CHUNKS=$nc IDG_TOPK_FORM=wave timeout 300 python scripts/eval_bench.py > gpurun_out/r2m/eval_wave_nc$nc.txt 2>&1; done
( time python bench.py > gpurun_out/r2m/bench_default.json 2> gpurun_out/r2m/bench_default.err ) 2> gpurun_out/r2m/bench_default.time
python __graft_entry__.py --smoke > gpurun_out/r2m/smoke.txt 2>&1; echo "smoke rc=$?" >> gpurun_out/r2m/smoke.txt
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" gpurun_out/r2m/topk_wave.txt | tail -n 6
grep -h "idg_score_topk 1 call\|identical" gpurun_out/r2m/eval_*.txt
cat gpurun_out/r2m/bench_default.time; tail -n 2 gpurun_out/r2m/smoke.txt; python scripts/brief.py < gpurun_out/r2m/bench_default.json
