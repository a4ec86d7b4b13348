#!/bin/bash
# Everything about the multi-GPU forms that ONE GPU can show (GPU box, from the repo root):
#   1. replicas (gradient-row exchange / dense all-reduce) and user-row shards at world size 1 through RCCL,
#      by the library's communicator and by torch.distributed;
#   2. the multi-rank bench rehearsed with 2/4/8 ranks sharing the GPU over gloo (host-staged collectives).
# Results: gpurun_out/multigpu/*.json
mkdir -p gpurun_out/multigpu
for comm in auto torch; do
  bash scripts/dp1.sh --steps 300 --warmup 30 --comm $comm > gpurun_out/multigpu/dp_rows_$comm.json 2> gpurun_out/multigpu/dp_rows_$comm.err
  bash scripts/dp1.sh --steps 300 --warmup 30 --comm $comm --dp-exchange grad > gpurun_out/multigpu/dp_grad_$comm.json 2> gpurun_out/multigpu/dp_grad_$comm.err
  bash scripts/sharded1.sh --steps 300 --warmup 30 --comm $comm > gpurun_out/multigpu/shard_$comm.json 2> gpurun_out/multigpu/shard_$comm.err
done
for n in 2 4 8; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2955$n \
    bench.py --gpus $n --backend gloo --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/multigpu/gloo_dp_$n.json 2> gpurun_out/multigpu/gloo_dp_$n.err
done
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29560 \
  bench.py --gpus 4 --backend gloo --parallel shard --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/multigpu/gloo_shard_4.json 2> gpurun_out/multigpu/gloo_shard_4.err
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/multigpu/*.json")):
    for l in open(f):
        if l.startswith("{"):
            j = json.loads(l)
            print("%-22s %8.4f ms/step  %12.0f triples/s  %s" % (f.split("/")[-1][:-5], j["ms_per_step"], j["value"], j["config"].get("comm")))
PY
