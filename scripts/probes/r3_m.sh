cd $GRAFT_REPO_ROOT
python -m cProfile -o /tmp/simgcl.prof bench.py --model SimGCL --workload amazon-book --batch 2048 --steps 400 --warmup 50 --no-cpu-baseline --epoch-leg off --hbm-leg off --scale-point off > /dev/null 2>&1
python - <<'PY'
import pstats
p = pstats.Stats("/tmp/simgcl.prof")
p.sort_stats("tottime").print_stats(28)
PY
