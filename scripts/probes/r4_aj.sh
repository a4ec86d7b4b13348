#!/bin/bash
# flakiness check of the paced engines: the model / parity suites three times, then three long epochs per model with loss checks
for i in 1 2 3; do python -m pytest tests/test_gpu_models.py tests/test_gpu_parity.py -x -q -p no:cacheprovider 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -1; done
for m in LightGCN SimGCL XSimGCL SGL EGCF NGCF MFBPR; do python scripts/e2e_epoch.py $m 6 2>&1 | grep "Training time" | tail -2 | cut -c1-120; done
