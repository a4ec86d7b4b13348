cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03_eval2 -o t -- python3 $R/scripts/eval_bench.py yelp2018 > $R/gpurun_out/prof_r03_eval2.log 2>&1
python3 - <<'PY'
import csv, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
rows=list(csv.DictReader(open(R+"/gpurun_out/prof_r03_eval2/t_kernel_trace.csv")))
agg=collections.defaultdict(list)
for r in rows:
    n=r["Kernel_Name"]
    if "score_topk" in n or "topk_merge" in n or "chunk_floor" in n:
        key=(n.split("(")[0][-60:], r["Grid_Size_X"], r["Grid_Size_Y"])
        agg[key].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in agg.items():
    v.sort(); print(k, len(v), "median %.1f us"%v[len(v)//2], "quartiles", [round(v[int(len(v)*q)]) for q in (0.05,0.25,0.45,0.55,0.75,0.95)])
PY
rm -f $R/gpurun_out/prof_r03_eval2/*kernel_trace.csv
