#!/bin/bash
mkdir -p gpurun_out/r2j
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "live_unit" > gpurun_out/r2j/units.txt 2>&1; echo "rc=$?" >> gpurun_out/r2j/units.txt
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r2j/pytest_all.txt 2>&1; echo "rc=$?" >> gpurun_out/r2j/pytest_all.txt
for u in 1 0; do
  IDG_LIVE_UNITS=$u timeout 300 python bench.py --no-cpu-baseline --hbm-leg off --epoch-leg off > gpurun_out/r2j/bench_units$u.json 2> gpurun_out/r2j/bench_units$u.err
  IDG_LIVE_UNITS=$u timeout 300 python bench.py --no-cpu-baseline --workload amazon-book --hbm-leg off --epoch-leg off > gpurun_out/r2j/bench_amazon_units$u.json 2> gpurun_out/r2j/bench_amazon_units$u.err
  IDG_LIVE_UNITS=$u timeout 300 python bench.py --no-cpu-baseline --workload amazon-book --model SimGCL --batch 2048 --hbm-leg off --epoch-leg off > gpurun_out/r2j/bench_simgcl_units$u.json 2> gpurun_out/r2j/bench_simgcl_units$u.err
done
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ngcf -o ngcf -- python3 $GRAFT_REPO_ROOT/scripts/e2e_epoch.py NGCF 2 > $GRAFT_REPO_ROOT/gpurun_out/r2j/prof_ngcf.log 2>&1; cd $GRAFT_REPO_ROOT
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" gpurun_out/r2j/units.txt | tail -n 12
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" gpurun_out/r2j/pytest_all.txt | tail -n 4
for t in units1 units0; do for f in bench_$t bench_amazon_$t bench_simgcl_$t; do echo "== $f"; python scripts/brief.py < gpurun_out/r2j/$f.json; done; done
head -14 gpurun_out/prof_ngcf/ngcf_kernel_stats.csv | cut -c1-160
