cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_models.py -x -q -m gpu -k "colsum or ngcf or wgrad" 2>&1 | tail -4
python scripts/e2e_epoch.py NGCF 3 2>&1 | grep -a "Training time" | tail -1
