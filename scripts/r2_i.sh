#!/bin/bash
mkdir -p gpurun_out/r2i
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r2i/pytest_all.txt 2>&1; echo "rc=$?" >> gpurun_out/r2i/pytest_all.txt
timeout 300 python scripts/e2e_epoch.py SGL 3 > gpurun_out/r2i/e2e_sgl.txt 2>&1
timeout 300 python scripts/e2e_epoch.py NGCF 3 > gpurun_out/r2i/e2e_ngcf.txt 2>&1
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" gpurun_out/r2i/pytest_all.txt | tail -n 12
grep "E2E\|Training time" gpurun_out/r2i/e2e_sgl.txt gpurun_out/r2i/e2e_ngcf.txt | tail -n 12
