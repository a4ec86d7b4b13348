cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_models.py tests/test_sharded.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do python bench.py --no-cpu-baseline --hbm-leg off --epoch-leg off --scale-point off --steps 1000 --warmup 100 2>/dev/null | python scripts/brief.py units_agg; done
bash scripts/sharded1.sh --workload yelp2018 --steps 300 --warmup 30 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('yelp shard1', d['ms_per_step'], d['host_issue_ms_per_step'])"
