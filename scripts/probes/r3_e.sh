cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in NGCF EGCF; do
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03_$m -o t -- python3 $R/scripts/e2e_epoch.py $m 3 > $R/gpurun_out/prof_r03_$m.log 2>&1
rm -f $R/gpurun_out/prof_r03_$m/*kernel_trace.csv
grep -a "Training time\|E2E" $R/gpurun_out/prof_r03_$m.log | tail -4
done
