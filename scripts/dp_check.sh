#!/bin/bash
mkdir -p gpurun_out/dp
for rep in 1 2 3; do
for comm in auto torch; do
bash scripts/sharded1.sh --steps 300 --warmup 30 --comm $comm > gpurun_out/dp/shard_${comm}_$rep.json 2> gpurun_out/dp/shard_${comm}_$rep.err
done
bash scripts/dp1.sh --steps 300 --warmup 30 > gpurun_out/dp/dp_auto_$rep.json 2> gpurun_out/dp/dp_auto_$rep.err
done
