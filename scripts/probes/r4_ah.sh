#!/bin/bash
python -m pytest tests/test_gpu_models.py tests/test_gpu_parity.py -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4
python bench.py --steps 1000 --warmup 100 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['value'], j['roofline']['hbm_bound']['frac'], j.get('epoch'))"
for m in EGCF NGCF LightGCN; do python scripts/e2e_epoch.py $m 4 2>&1 | grep "Training time" | tail -1; done
