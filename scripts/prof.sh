#!/bin/bash
# usage (on the GPU box, from the repo root): bash scripts_prof.sh <tag> [bench args...]
# kernel-trace + stats only (no PMC here; PMC passes are separate runs)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -o $tag -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --hbm-leg off --epoch-leg off "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
