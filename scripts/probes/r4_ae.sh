#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for m in EGCF SimGCL; do
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r04g_$m -o $m -- python3 $GRAFT_REPO_ROOT/scripts/e2e_epoch.py $m 3 > $GRAFT_REPO_ROOT/gpurun_out/prof_r04g_$m.log 2>&1
grep -a "Training time" $GRAFT_REPO_ROOT/gpurun_out/prof_r04g_$m.log | tail -1
done
