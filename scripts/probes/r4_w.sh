#!/bin/bash
for wgs in 768 1024 1536 2048 512; do echo "== WGS $wgs"; IDG_NGCF_WGS=$wgs python scripts/probes/ngcf_transform_bench.py 2>&1 | grep " us"; done
for pr in 1 3; do echo "== probe $pr"; IDG_NGCF_PROBE=$pr python scripts/probes/ngcf_transform_bench.py 2>&1 | grep "fwd"; done
