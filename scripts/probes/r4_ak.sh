#!/bin/bash
B="--no-cpu-baseline --scale-point off --hbm-leg off --epoch-leg off --steps 2000 --warmup 100"
for i in 1 2 3; do
echo -n "new : "; python bench.py $B 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['loss_last'] if 'loss_last' in j else '')"
echo -n "base: "; IDG_LIB_PATH=$PWD/id-grec_amd/lib_base/libidgrec.so python bench.py $B 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['loss_last'] if 'loss_last' in j else '')"
done
python -m pytest tests/test_gpu_parity.py -x -q -k "restricted or units or live_unit or out_rows or engine or trajectory" 2>&1 | tail -2
