#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "receptive" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tail -n 25
