#!/bin/bash
python -m pytest tests/test_gpu_parity.py tests/test_gpu_models.py -x -q -k "infonce or ssl or simgcl or sgl or egcf or lookahead" 2>&1 | tail -3
for m in EGCF SimGCL XSimGCL SGL; do python scripts/e2e_epoch.py $m 4 2>&1 | grep "Training time" | tail -1; done
