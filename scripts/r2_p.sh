#!/bin/bash
mkdir -p gpurun_out/r2p
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/r2p/pytest_all.txt 2>&1; echo "rc=$?" >> gpurun_out/r2p/pytest_all.txt
python __graft_entry__.py --smoke > gpurun_out/r2p/smoke.txt 2>&1; echo "smoke rc=$?" >> gpurun_out/r2p/smoke.txt
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29562 \
  bench.py --gpus 2 --backend gloo --workload synth-1M --dim 256 --steps 4 --warmup 2 --cpu-seconds 4 > gpurun_out/r2p/gloo2_synth1m_d256.json 2> gpurun_out/r2p/gloo2_synth1m_d256.err
bash scripts/sharded1.sh --workload yelp2018 --steps 300 --warmup 30 > gpurun_out/r2p/shard1_yelp.json 2> gpurun_out/r2p/shard1_yelp.err
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" gpurun_out/r2p/pytest_all.txt | tail -n 6; tail -n 2 gpurun_out/r2p/smoke.txt
cut -c1-900 gpurun_out/r2p/gloo2_synth1m_d256.json; tail -n 3 gpurun_out/r2p/gloo2_synth1m_d256.err; python scripts/brief.py < gpurun_out/r2p/shard1_yelp.json
