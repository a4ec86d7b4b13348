#!/bin/bash
mkdir -p gpurun_out/r2q
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/r2q/pytest_all.txt 2>&1; echo "rc=$?" >> gpurun_out/r2q/pytest_all.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r2q/bench_driver_form.json 2> gpurun_out/r2q/bench_driver_form.err
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" gpurun_out/r2q/pytest_all.txt | tail -n 5
python scripts/brief.py < gpurun_out/r2q/bench_driver_form.json
