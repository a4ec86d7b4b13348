#!/bin/bash
python -m pytest tests/test_gpu_parity.py -x -q -k "ngcf_layer_kernels" 2>&1 | tail -2
python scripts/probes/ngcf_transform_bench.py 2>&1 | grep "LAYER"
python scripts/e2e_epoch.py NGCF 4 2>&1 | grep "Training time" | tail -1
