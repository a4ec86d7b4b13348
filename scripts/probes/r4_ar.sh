#!/bin/bash
python -m pytest tests/test_gpu_models.py -x -q -k "xsimgcl or lookahead" 2>&1 | tail -2
python scripts/e2e_epoch.py XSimGCL 5 2>&1 | grep "Training time" | tail -2 | cut -c1-120
