cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "egcf" 2>&1 | grep -a "passed\|failed" | tail -2
python scripts/e2e_epoch.py EGCF 3 2>&1 | grep -a "Training time" | tail -2
