#!/bin/bash
bash scripts/prof.sh simgcl --workload amazon-book --model SimGCL --steps 100
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_simgcl/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:22]:
    print("%-110s %6s %10.1f %6s" % (r['Name'][:110], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
tail -2 gpurun_out/prof_simgcl.log | cut -c1-300
rm -rf gpurun_out/prof_simgcl
