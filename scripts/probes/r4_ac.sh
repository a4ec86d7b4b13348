#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r04f_ssl -o ssl -- python3 $GRAFT_REPO_ROOT/scripts/probes/infonce_ab.py > $GRAFT_REPO_ROOT/gpurun_out/prof_r04f_ssl.log 2>&1
grep " us" $GRAFT_REPO_ROOT/gpurun_out/prof_r04f_ssl.log
