#!/bin/bash
# lookahead preparation for the EGCF / NGCF fused steps: parity tests + epoch time
mkdir -p gpurun_out/r4u
python -m pytest tests/test_gpu_models.py -x -q  > gpurun_out/r4u/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r4u/pytest.txt
tail -3 gpurun_out/r4u/pytest.txt
python scripts/e2e_epoch.py EGCF 4 2>&1 | grep "Training time" | tail -2
python scripts/e2e_epoch.py NGCF 4 2>&1 | grep "Training time" | tail -2
