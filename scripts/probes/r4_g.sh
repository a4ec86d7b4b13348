cd $GRAFT_REPO_ROOT
IDG_COMPACT_INPUTS=1 bash scripts/prof.sh r04_compact1 --scale-point off
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_r04_compact1/r04_compact1_kernel_stats.csv')))
for r in rows[:16]:
    print("%-90s %6s %9.1f us"%(r['Name'].replace('(anonymous namespace)::','')[:90], r['Calls'], float(r['AverageNs'])/1e3))
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r04_ngcf -o ngcf -- python3 $GRAFT_REPO_ROOT/scripts/e2e_epoch.py NGCF 3 > $GRAFT_REPO_ROOT/gpurun_out/prof_r04_ngcf.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_r04_ngcf/ngcf_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("NGCF total %.1f ms"%(tot/1e6))
for r in rows[:22]:
    print("%-90s %6s %9.1f us %5.1f%%"%(r['Name'].replace('(anonymous namespace)::','')[:90], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/tot*100))
PY
