#!/bin/bash
mkdir -p gpurun_out/r2b
timeout 600 python -m pytest tests/test_gpu_scale.py tests/test_gpu_models.py -x -q -m gpu -k "simgcl_encoder or without_tiled" > gpurun_out/r2b/fixed.txt 2>&1; echo "rc=$?" >> gpurun_out/r2b/fixed.txt
timeout 300 python scripts/nt_probe.py yelp2018 64 > gpurun_out/r2b/nt_yelp.txt 2>&1
timeout 300 python scripts/nt_probe.py amazon-book 64 0,8000,16000,32000,64000 > gpurun_out/r2b/nt_amazon.txt 2>&1
timeout 600 python scripts/nt_probe.py synth-1M 64 0,16000,64000,128000,256000,512000 > gpurun_out/r2b/nt_synth1m.txt 2>&1
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" gpurun_out/r2b/fixed.txt | tail -n 15; cat gpurun_out/r2b/nt_*.txt
