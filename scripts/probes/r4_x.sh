#!/bin/bash
# fused NGCF layer kernels: parity vs the chain, the fused step's tests, epoch time
mkdir -p gpurun_out/r4x
python -m pytest tests/test_gpu_parity.py -x -q -k "ngcf_layer_kernels" > gpurun_out/r4x/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r4x/pytest.txt
python -m pytest tests/test_gpu_models.py -x -q -k "ngcf or lookahead" >> gpurun_out/r4x/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r4x/pytest.txt
grep -v "^Extension\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" gpurun_out/r4x/pytest.txt | tail -25
python scripts/e2e_epoch.py NGCF 4 2>&1 | grep "Training time" | tail -2
IDG_NGCF_LAYER=0 python scripts/e2e_epoch.py NGCF 4 2>&1 | grep "Training time" | tail -1
