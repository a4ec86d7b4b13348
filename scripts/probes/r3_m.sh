cd $GRAFT_REPO_ROOT
python -m cProfile -s tottime bench.py --model SimGCL --workload amazon-book --batch 2048 --steps 400 --warmup 50 --no-cpu-baseline --epoch-leg off --hbm-leg off --scale-point off 2>/dev/null | head -45
