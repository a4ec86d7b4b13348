cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "live_unit or two_row_sets or epilogue_struct or row_movers or bpr" 2>&1 | tail -8
python -m pytest tests/test_sharded.py tests/test_bench_contract.py -x -q -m gpu 2>&1 | tail -8
python -m pytest tests/test_gpu_scale.py -x -q -m gpu -k "sharded_step_at_config5" 2>&1 | tail -5
bash scripts/sharded1.sh --workload yelp2018 --steps 300 --warmup 30 > gpurun_out/r03_shard1_yelp.json 2> gpurun_out/r03_shard1_yelp.err; python -c "
import json; d=json.load(open('gpurun_out/r03_shard1_yelp.json')); print('yelp shard1', d['ms_per_step'], d['host_issue_ms_per_step'])"
bash scripts/sharded1.sh --workload synth-10M --dim 256 --steps 6 --warmup 3 > gpurun_out/r03_shard1_c5.json 2> gpurun_out/r03_shard1_c5.err; python -c "
import json; d=json.load(open('gpurun_out/r03_shard1_c5.json')); print('c5 shard1', d['ms_per_step'], d['host_issue_ms_per_step'])"
tail -3 gpurun_out/r03_shard1_c5.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r03_eval -o t -- python3 $GRAFT_REPO_ROOT/scripts/eval_bench.py yelp2018 > $GRAFT_REPO_ROOT/gpurun_out/prof_r03_eval.log 2>&1
grep "ms per full" $GRAFT_REPO_ROOT/gpurun_out/prof_r03_eval.log
