#!/bin/bash
mkdir -p gpurun_out/dp
python bench.py --no-cpu-baseline --steps 50 --warmup 5 > gpurun_out/dp/o1.out 2> gpurun_out/dp/o1.err
bash scripts/dp1.sh --steps 50 --warmup 5 > gpurun_out/dp/o2.out 2> gpurun_out/dp/o2.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29557 bench.py --gpus 2 --backend gloo --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/dp/o3.out 2> gpurun_out/dp/o3.err
wc -l gpurun_out/dp/o?.out; head -c 150 gpurun_out/dp/o2.out; echo; grep -c "RCCL version" gpurun_out/dp/o2.err
