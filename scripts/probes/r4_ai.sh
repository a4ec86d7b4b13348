#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for sl in 1 2 3 4; do
export IDG_TOPK_FLOOR_SLABS_RT=$sl
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r04i_sl$sl -o t -- python3 $GRAFT_REPO_ROOT/scripts/probes/topk_startup_probe.py > /dev/null 2>&1
echo "slabs $sl"; grep "score_topk_spec\|chunk_floor\|topk_merge" $GRAFT_REPO_ROOT/gpurun_out/prof_r04i_sl$sl/t_kernel_stats.csv | cut -d, -f1-5 | cut -c1-60,150-
done
