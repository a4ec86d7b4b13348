#!/bin/bash
python -m pytest tests/test_gpu_models.py -x -q -k "egcf or lookahead" 2>&1 | tail -2
E2E_CONFIG=mode=alternating python scripts/e2e_epoch.py EGCF 4 2>&1 | grep "Training time" | tail -1
python scripts/e2e_epoch.py EGCF 4 2>&1 | grep "Training time" | tail -1
