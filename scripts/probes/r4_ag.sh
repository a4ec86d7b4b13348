#!/bin/bash
for i in 1 2 3; do
for v in 0 1; do
echo -n "IDG_PACE=$v: "; IDG_PACE=$v python bench.py --no-cpu-baseline --scale-point off --hbm-leg off --epoch-leg off --steps 2000 --warmup 100 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"
done; done
