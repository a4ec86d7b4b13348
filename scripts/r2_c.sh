#!/bin/bash
# round 2, third GPU session: the re-written user-row-sharded step (guest rows, sliced all-reduces, sharded evaluation)
mkdir -p gpurun_out/r2c
timeout 900 python -m pytest tests/test_sharded.py tests/test_replicated.py -x -q -m gpu > gpurun_out/r2c/dist_tests.txt 2>&1; echo "rc=$?" >> gpurun_out/r2c/dist_tests.txt
# world 1 through RCCL (library communicator): yelp2018 shape, sharded form and replica form
bash scripts/sharded1.sh --workload yelp2018 --steps 300 --warmup 30 > gpurun_out/r2c/shard1_yelp.json 2> gpurun_out/r2c/shard1_yelp.err
bash scripts/dp1.sh --workload yelp2018 --steps 300 --warmup 30 > gpurun_out/r2c/dp1_yelp.json 2> gpurun_out/r2c/dp1_yelp.err
# two ranks sharing the GPU over gloo: the default multi-GPU line (sharded headline + replicas field), yelp2018 shape
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29552 \
  bench.py --gpus 2 --backend gloo --workload yelp2018 --steps 30 --warmup 5 --cpu-seconds 5 > gpurun_out/r2c/gloo2_default.json 2> gpurun_out/r2c/gloo2_default.err
# config-5 size through the sharded path at world 1 (one rank holds the whole 10M-user graph)
bash scripts/sharded1.sh --workload synth-10M --dim 256 --steps 8 --warmup 3 > gpurun_out/r2c/shard1_c5.json 2> gpurun_out/r2c/shard1_c5.err
timeout 300 python scripts/l2_probe.py yelp2018 64 > gpurun_out/r2c/l2_probe_yelp.txt 2>&1
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" gpurun_out/r2c/dist_tests.txt | tail -n 12
for f in shard1_yelp dp1_yelp gloo2_default shard1_c5; do echo "== $f"; cat gpurun_out/r2c/$f.json; tail -n 3 gpurun_out/r2c/$f.err; done
cat gpurun_out/r2c/l2_probe_yelp.txt
