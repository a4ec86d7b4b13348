#!/bin/bash
# world-size-1 run of the data-parallel replicas (exercises the RCCL code path on one GPU): bash scripts/dp1.sh [bench args]
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=${MASTER_PORT:-29513}
python bench.py --gpus 1 --force-sharded --parallel dp --no-cpu-baseline "$@"
