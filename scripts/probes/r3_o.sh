cd $GRAFT_REPO_ROOT
python -m pytest tests/test_sharded.py tests/test_replicated.py tests/test_bench_contract.py -x -q -m gpu 2>&1 | tail -3
python -m pytest tests/test_gpu_scale.py -x -q -m gpu -k "sharded_step_at_config5" 2>&1 | tail -2
for i in 1 2; do bash scripts/sharded1.sh --workload yelp2018 --steps 300 --warmup 30 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('yelp shard1', d['ms_per_step'], d['host_issue_ms_per_step'])"; done
bash scripts/sharded1.sh --workload synth-10M --dim 256 --steps 6 --warmup 3 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('c5 shard1', d['ms_per_step'], d['host_issue_ms_per_step'])"
