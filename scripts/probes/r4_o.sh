# round 4, session O: the DEFAULT multi-GPU line (configs[4], user-row shards) rehearsed with two ranks sharing the GPU over gloo
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4o
( time timeout 1700 python bench.py --gpus 2 --backend gloo --steps 2 --warmup 1 --no-cpu-baseline ) > gpurun_out/r4o/shard_gloo2_c5.json 2> gpurun_out/r4o/shard_gloo2_c5.err
tail -5 gpurun_out/r4o/shard_gloo2_c5.err
python - <<'PY'
import json
for l in open('gpurun_out/r4o/shard_gloo2_c5.json'):
    if l.startswith('{'):
        d=json.loads(l)
        print({k:d.get(k) for k in ('value','ms_per_step','n_gpus','item_table_coherent','speedup_vs_1gpu','replicated_bytes_per_step_per_rank','error','retried')})
        t=d.get('timeline') or {}
        print({k:t.get(k) for k in ('step_gpu_ms','compute_ms','host_sync_ms','touched_item_rows_exchanged','touched_item_agreement')})
        print(d['config']['workload'][:400])
        print(d.get('single_gpu_reference',{}).get('ms_per_step'))
PY
