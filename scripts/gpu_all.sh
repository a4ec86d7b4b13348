#!/bin/bash
# whole GPU suite + smoke + the default bench, outputs under gpurun_out/all/
mkdir -p gpurun_out/all
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/all/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/all/pytest.txt
python __graft_entry__.py --smoke > gpurun_out/all/smoke.txt 2>&1; echo "smoke rc=$?" >> gpurun_out/all/smoke.txt
python bench.py > gpurun_out/all/bench.json 2> gpurun_out/all/bench.err
for i in 1 2; do bash scripts/sharded1.sh --steps 300 --warmup 30 > gpurun_out/all/shard_$i.json 2> gpurun_out/all/shard_$i.err; done
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" gpurun_out/all/pytest.txt | tail -5; tail -2 gpurun_out/all/smoke.txt
