cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_scale.py -q -m gpu -rs 2>&1 | tail -6
