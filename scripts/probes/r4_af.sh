#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r04h_mf -o mf -- python3 $GRAFT_REPO_ROOT/bench.py --model MFBPR --batch 2048 --no-cpu-baseline --scale-point off --hbm-leg off --epoch-leg off --steps 500 > $GRAFT_REPO_ROOT/gpurun_out/prof_r04h_mf.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_r04h_mf.log | cut -c1-300
