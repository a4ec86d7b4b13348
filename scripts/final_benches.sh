#!/bin/bash
# end-of-round bench lines for every BASELINE config that fits one GPU (GPU box, repo root) -> gpurun_out/final/
mkdir -p gpurun_out/final
python bench.py --workload amazon-book --no-cpu-baseline > gpurun_out/final/amazon.json 2> gpurun_out/final/amazon.err
python bench.py --workload synth-1M --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/final/synth1m.json 2> gpurun_out/final/synth1m.err
python bench.py --workload amazon-book --model SimGCL --batch 2048 --steps 400 --warmup 40 --no-cpu-baseline > gpurun_out/final/simgcl_amazon.json 2> gpurun_out/final/simgcl_amazon.err
python bench.py --model MFBPR --batch 2048 --no-cpu-baseline > gpurun_out/final/mfbpr.json 2> gpurun_out/final/mfbpr.err
python bench.py --workload synth-10M --dim 256 --steps 15 --warmup 4 --no-cpu-baseline > gpurun_out/final/c5.json 2> gpurun_out/final/c5.err
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/final/*.json")):
    for l in open(f):
        if l.startswith("{"):
            j = json.loads(l)
            print("%-16s %9.4f ms/step %12.0f %s  dense %.1f us" % (f.split("/")[-1][:-5], j["ms_per_step"], j["value"], j["unit"], (j.get("roofline") or {}).get("us_per_launch") or 0))
PY
