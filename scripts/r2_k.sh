#!/bin/bash
mkdir -p gpurun_out/r2k
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "live_unit or topk or score" > gpurun_out/r2k/units_topk.txt 2>&1; echo "rc=$?" >> gpurun_out/r2k/units_topk.txt
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r2k/pytest_all.txt 2>&1; echo "rc=$?" >> gpurun_out/r2k/pytest_all.txt
for form in wave slab; do
  IDG_TOPK_FORM=$form timeout 300 python scripts/eval_bench.py > gpurun_out/r2k/eval_$form.txt 2>&1
  IDG_TOPK_FORM=$form timeout 300 python scripts/eval_bench.py amazon-book > gpurun_out/r2k/eval_amazon_$form.txt 2>&1
done
timeout 300 python bench.py --no-cpu-baseline --hbm-leg off --epoch-leg off > gpurun_out/r2k/bench.json 2> gpurun_out/r2k/bench.err
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" gpurun_out/r2k/units_topk.txt | tail -n 25
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" gpurun_out/r2k/pytest_all.txt | tail -n 12
tail -n 4 gpurun_out/r2k/eval_*.txt
python scripts/brief.py < gpurun_out/r2k/bench.json
