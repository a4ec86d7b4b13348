cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "topk" 2>&1 | tail -5
python scripts/eval_bench.py yelp2018 2>&1 | grep "ms per full\|one call\|identical"
IDG_TOPK_FLOOR=0 python scripts/eval_bench.py yelp2018 2>&1 | grep "ms per full"
python scripts/eval_bench.py amazon-book 2>&1 | grep "ms per full\|one call\|identical"
