# full GPU suite + smoke + the default bench line + the 2-rank gloo rehearsal of the configs[4] line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3g
( time python -m pytest tests -q -m gpu -x ) > gpurun_out/r3g/pytest_gpu.log 2>&1; tail -4 gpurun_out/r3g/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time python bench.py ) > gpurun_out/r3g/bench_default.json 2> gpurun_out/r3g/bench_default.err; tail -3 gpurun_out/r3g/bench_default.err; python scripts/brief.py default < gpurun_out/r3g/bench_default.json
python -c "
import json
d=json.load(open('gpurun_out/r3g/bench_default.json'))
print({k:d['roofline'][k] for k in ('bound','frac','ceiling','frac_of_ceiling','us_per_launch')})
print('hbm_bound', {k:d['roofline']['hbm_bound'][k] for k in ('frac','ceiling','frac_of_ceiling','us_per_launch','bound')})
print('scale_point', d.get('scale_point'))
print('cpu', d['cpu_baseline']['value'], 'epoch', d['epoch']['epoch_s'], 'eval', d['eval']['ms_per_full_evaluation'])
"
( time python bench.py --gpus 2 --backend gloo --steps 2 --warmup 1 --no-cpu-baseline ) > gpurun_out/r3g/shard_gloo2_c5.json 2> gpurun_out/r3g/shard_gloo2_c5.err; tail -3 gpurun_out/r3g/shard_gloo2_c5.err
python -c "
import json
d=json.load(open('gpurun_out/r3g/shard_gloo2_c5.json'))
print(d['ms_per_step'], d['item_table_coherent'], d.get('speedup_vs_1gpu'), d['single_gpu_reference'], d['replicated_bytes_per_step_per_rank'], d['roofline']['exchange_rows'])
"
