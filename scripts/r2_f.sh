#!/bin/bash
mkdir -p gpurun_out/r2f
run() { # tag
  timeout 300 python bench.py --no-cpu-baseline --hbm-leg off --epoch-leg off > gpurun_out/r2f/bench_$1.json 2> gpurun_out/r2f/bench_$1.err
  timeout 300 python bench.py --no-cpu-baseline --workload amazon-book --hbm-leg off --epoch-leg off > gpurun_out/r2f/bench_amazon_$1.json 2> gpurun_out/r2f/bench_amazon_$1.err
  timeout 300 python bench.py --no-cpu-baseline --workload amazon-book --model SimGCL --batch 2048 --hbm-leg off --epoch-leg off > gpurun_out/r2f/bench_simgcl_$1.json 2> gpurun_out/r2f/bench_simgcl_$1.err
}
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/r2f/pytest_all.txt 2>&1; echo "rc=$?" >> gpurun_out/r2f/pytest_all.txt
run u8
for u in 16 32; do
  IDG_BUILD_DEFS="-DIDG_ROWS_UNROLL=$u" python id-grec_amd/build.py --force > gpurun_out/r2f/build_u$u.log 2>&1
  run u$u
done
python id-grec_amd/build.py --force > gpurun_out/r2f/build_default.log 2>&1
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" gpurun_out/r2f/pytest_all.txt | tail -n 8
for t in u8 u16 u32; do for f in bench_$t bench_amazon_$t bench_simgcl_$t; do echo "== $f"; python scripts/brief.py < gpurun_out/r2f/$f.json; done; done
