cd $GRAFT_REPO_ROOT
python scripts/probes/topk_floor_debug.py 2>&1 | grep IDG_TOPK
cd /tmp && export TMPDIR=/tmp
IDG_TOPK_FLOOR=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r04_topk1 -o t -- python3 $GRAFT_REPO_ROOT/scripts/probes/topk_floor_debug.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_r04_topk1/t_kernel_stats.csv')))
for r in rows[:8]:
    print("%-100s %6s avg %9.1f us min %9.1f max %9.1f"%(r['Name'].replace('(anonymous namespace)::','')[:100], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
