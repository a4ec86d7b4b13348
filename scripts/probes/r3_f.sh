cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "live_unit or restricted" 2>&1 | tail -3
for i in 1 2; do python bench.py --no-cpu-baseline --hbm-leg off --epoch-leg off --scale-point off --steps 1000 --warmup 100 2>/dev/null | python scripts/brief.py prefetch; done
IDG_BUILD_DEFS="-DIDG_UNITS_PREFETCH=0" python id-grec_amd/build.py --force > /dev/null 2>&1
for i in 1 2; do python bench.py --no-cpu-baseline --hbm-leg off --epoch-leg off --scale-point off --steps 1000 --warmup 100 2>/dev/null | python scripts/brief.py noprefetch; done
