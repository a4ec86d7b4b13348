cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "live_unit or two_row_sets or epilogue_struct or row_movers" 2>&1 | tail -15
python -m pytest tests/test_sharded.py -x -q -m gpu 2>&1 | tail -15
python -m pytest tests/test_gpu_scale.py -x -q -m gpu -k "sharded_step_at_config5" 2>&1 | tail -15
bash scripts/sharded1.sh --workload yelp2018 --steps 300 --warmup 30 > gpurun_out/r03_shard1_yelp.json 2> gpurun_out/r03_shard1_yelp.err; tail -c 1500 gpurun_out/r03_shard1_yelp.json; tail -5 gpurun_out/r03_shard1_yelp.err
bash scripts/sharded1.sh --workload synth-10M --dim 256 --steps 6 --warmup 3 > gpurun_out/r03_shard1_c5.json 2> gpurun_out/r03_shard1_c5.err; tail -c 2500 gpurun_out/r03_shard1_c5.json; tail -5 gpurun_out/r03_shard1_c5.err
