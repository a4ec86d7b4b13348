#!/bin/bash
# The sharded step rehearsed with 2 ranks sharing the GPU over gloo (host-staged collectives: the code path of an N-rank
# run, NOT a performance number) — bench.py starts its own ranks.  usage: bash scripts/rehearse_shard2.sh [bench args]
mkdir -p gpurun_out/reh
timeout 1500 python bench.py --gpus 2 --backend gloo --parallel shard --workload ${WORKLOAD:-synth-1M} --dim ${DIM:-256} \
  --steps 6 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/reh/shard2.json 2> gpurun_out/reh/shard2.err
python - gpurun_out/reh/shard2.json <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print("%.1f ms/step" % d["ms_per_step"], d["roofline"]["exchange_rows"], "coherent", d["item_table_coherent"],
              "%.0f MB sent/step/rank" % (d["roofline"]["exchange_bytes_per_step_per_rank"] / 1e6), d["loss_last"],
              "speedup_vs_1gpu", d.get("speedup_vs_1gpu"))
PY
