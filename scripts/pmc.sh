#!/bin/bash
# usage (GPU box, repo root): bash scripts/pmc.sh <tag> [bench args...]
# Separate PMC passes (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2: cannot share a pass), kernel-trace only.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_$tag
for ctr in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  name=$(echo $ctr | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $R/gpurun_out/pmc_$tag/$name -o p -- python3 $R/bench.py --no-cpu-baseline --hbm-leg off --epoch-leg off --steps 20 --warmup 5 "$@" > $R/gpurun_out/pmc_$tag/$name.log 2>&1 || echo "pass $name failed"
done
python3 $R/scripts/pmc_summary.py $R/gpurun_out/pmc_$tag > $R/gpurun_out/pmc_$tag/summary.json
cat $R/gpurun_out/pmc_$tag/summary.json
