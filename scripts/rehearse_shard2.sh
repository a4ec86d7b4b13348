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
        t = d.get("timeline") or {}
        print("timeline: step %.2f ms on the GPU, stalls %.2f, compute %.2f, host sync %.3f ms; %s"
              % (t.get("step_gpu_ms", 0), t.get("main_stream_stall_ms", 0), t.get("compute_ms", 0), t.get("host_sync_ms", 0),
                 t.get("touched_item_agreement")))
        for tag, v in sorted((t.get("collectives") or {}).items()):
            print("  %-24s %-14s x%.0f  %8.2f MB  %7.3f ms  stall %7.3f ms  bus %s GB/s"
                  % (tag, v["kind"], v["calls_per_step"], v["bytes_per_step"] / 1e6, v["collective_ms"],
                     v["main_stream_stall_ms"], ("%.1f" % v["bus_gbs"]) if v["bus_gbs"] else "-"))
        if d.get("retried"):
            print("RETRIED:", d["retried"]["why"])
PY
