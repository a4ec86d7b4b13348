cd /tmp && export TMPDIR=/tmp
for m in EGCF; do
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r04b_$m -o $m -- python3 $GRAFT_REPO_ROOT/scripts/e2e_epoch.py $m 3 > $GRAFT_REPO_ROOT/gpurun_out/prof_r04b_$m.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv
for m in ("EGCF",):
    rows=list(csv.DictReader(open('gpurun_out/prof_r04b_%s/%s_kernel_stats.csv'%(m,m))))
    tot=sum(float(r['TotalDurationNs']) for r in rows)
    print(m,"total %.0f ms"%(tot/1e6))
    for r in rows[:12]:
        print("   %-80s %6s %8.1f us %5.1f%%"%(r['Name'].replace('(anonymous namespace)::','')[:80], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/tot*100))
PY
