# PMC passes where the dense product is HBM-bound: synth-10M at d=64 (3.84 GB panel) and d=256 (15.4 GB panel)
cd $GRAFT_REPO_ROOT
bash scripts/pmc.sh r03_synth10M_d64 --workload synth-10M --dim 64 --separate-adam --scale-point off > gpurun_out/pmc_r03_synth10M_d64.out 2>&1
python scripts/traffic_json.py gpurun_out/pmc_r03_synth10M_d64/summary.json synth-10M 64 r03 | tail -25
bash scripts/pmc.sh r03_synth10M_d256 --workload synth-10M --dim 256 --separate-adam --scale-point off > gpurun_out/pmc_r03_synth10M_d256.out 2>&1
python scripts/traffic_json.py gpurun_out/pmc_r03_synth10M_d256/summary.json synth-10M 256 r03 | tail -25
cp profiles/r03/traffic_synth-10M_d*.json gpurun_out/ 2>/dev/null
