#!/bin/bash
B="--no-cpu-baseline --scale-point off --hbm-leg off --epoch-leg off --steps 2000 --warmup 100"
for i in 1 2 3; do
for v in 0 1; do
echo -n "store_grad=$v: "; IDG_BENCH_STORE_GRAD=$v python bench.py $B 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['loss_first_last'])"
done; done
python -m pytest tests/test_gpu_parity.py tests/test_gpu_models.py -x -q 2>&1 | tail -2
