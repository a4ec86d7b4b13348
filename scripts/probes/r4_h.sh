cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4h
timeout 1200 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "ngcf" > gpurun_out/r4h/pytest.txt 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r4h/pytest.txt
python scripts/e2e_epoch.py NGCF 4 2>&1 | grep -a "Training time\|Error\|error" | tail -3
