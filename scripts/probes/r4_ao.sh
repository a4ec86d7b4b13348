#!/bin/bash
cd /tmp && export TMPDIR=/tmp
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 IDG_BENCH_SUPERVISE=0
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r04_sh1 -o sh1 -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --force-sharded --parallel shard --steps 300 --warmup 30 --no-cpu-baseline --scale-point off > $GRAFT_REPO_ROOT/gpurun_out/prof_r04_sh1.log 2>&1
tail -2 $GRAFT_REPO_ROOT/gpurun_out/prof_r04_sh1.log | cut -c1-300
