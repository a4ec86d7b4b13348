#!/bin/bash
for w in 512 768 1024 1280 1536 2048; do
echo "wgs=$w"; IDG_TOPK_WGS=$w python scripts/topk_only.py yelp2018 5 2>&1 | grep evaluation; IDG_TOPK_WGS=$w python scripts/topk_only.py amazon-book 3 2>&1 | grep evaluation
done
