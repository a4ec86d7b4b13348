#!/bin/bash
# top-K: tiled item operand + batched merge loads: evaluation time in calls of 1024 users / one call
python scripts/eval_bench.py yelp2018 2>&1 | grep "ms per full"
python scripts/eval_bench.py amazon-book 2>&1 | grep "ms per full"
