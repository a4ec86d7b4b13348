#!/bin/bash
# world-size-1 run of the user-row-sharded path (exercises the RCCL code path on one GPU): bash scripts/sharded1.sh [bench args]
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=${MASTER_PORT:-29512}
python bench.py --gpus 1 --force-sharded --parallel shard --no-cpu-baseline "$@"
