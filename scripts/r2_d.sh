#!/bin/bash
mkdir -p gpurun_out/r2d
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r2d/parity.txt 2>&1; echo "rc=$?" >> gpurun_out/r2d/parity.txt
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_scale.py -x -q -m gpu > gpurun_out/r2d/models.txt 2>&1; echo "rc=$?" >> gpurun_out/r2d/models.txt
for srt in 1 0; do
  IDG_TILE_SORT=$srt timeout 300 python bench.py --no-cpu-baseline --hbm-leg on --epoch-leg off > gpurun_out/r2d/bench_sort$srt.json 2> gpurun_out/r2d/bench_sort$srt.err
  IDG_TILE_SORT=$srt timeout 300 python bench.py --no-cpu-baseline --workload amazon-book --hbm-leg off --epoch-leg off > gpurun_out/r2d/bench_amazon_sort$srt.json 2> gpurun_out/r2d/bench_amazon_sort$srt.err
done
IDG_TILE_SORT=1 timeout 300 python scripts/l2_probe.py yelp2018 64 > gpurun_out/r2d/l2_probe_sorted.txt 2>&1
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" gpurun_out/r2d/parity.txt | tail -n 15
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" gpurun_out/r2d/models.txt | tail -n 15
for f in bench_sort1 bench_sort0 bench_amazon_sort1 bench_amazon_sort0; do echo "== $f"; python scripts/brief.py < gpurun_out/r2d/$f.json; tail -n 2 gpurun_out/r2d/$f.err; done
cat gpurun_out/r2d/l2_probe_sorted.txt
