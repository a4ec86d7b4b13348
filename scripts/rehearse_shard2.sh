#!/bin/bash
# The sharded step rehearsed with 2 ranks sharing the GPU over gloo (host-staged collectives) on synth-1M at d = 256:
# the compact two-hop exchanges against the panel ones (IDG_TWO_HOP=0: forward layer K-2 / second backward product as
# panels).  At B = 1024 the two-hop items of this 500 k-item graph are most of the table (the exchange falls back to the
# panel by itself); B = 64 is the regime configs[4] is in at B = 1024 (10x the rows, the same batch).
mkdir -p gpurun_out/reh
for b in ${BATCHES:-64 1024}; do for th in 1 0; do
  IDG_TWO_HOP=$th timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2957$th \
    bench.py --gpus 2 --backend gloo --parallel shard --workload synth-1M --dim 256 --batch $b --steps 10 --warmup 3 --no-cpu-baseline \
    > gpurun_out/reh/shard2_b${b}_twohop$th.json 2> gpurun_out/reh/shard2_b${b}_twohop$th.err
  python - gpurun_out/reh/shard2_b${b}_twohop$th.json <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print(sys.argv[1].split("/")[-1], "%.1f ms/step" % d["ms_per_step"], d["roofline"]["exchange_rows"],
              "%.0f MB exchanged/step/rank" % (d["roofline"]["exchange_bytes_per_step_per_rank"] / 1e6), d["loss_last"])
PY
done; done
