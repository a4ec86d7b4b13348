#!/bin/bash
mkdir -p gpurun_out/r2g
run() { # tag
  timeout 300 python bench.py --no-cpu-baseline --hbm-leg on --epoch-leg off > gpurun_out/r2g/bench_$1.json 2> gpurun_out/r2g/bench_$1.err
  timeout 300 python bench.py --no-cpu-baseline --workload amazon-book --hbm-leg off --epoch-leg off > gpurun_out/r2g/bench_amazon_$1.json 2> gpurun_out/r2g/bench_amazon_$1.err
}
run minw1
IDG_BUILD_DEFS="-DIDG_FUSED_MINW=8" python id-grec_amd/build.py --force > gpurun_out/r2g/build_minw8.log 2>&1
run minw8
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r2g/parity_minw8.txt 2>&1; echo "rc=$?" >> gpurun_out/r2g/parity_minw8.txt
python id-grec_amd/build.py --force > gpurun_out/r2g/build_default.log 2>&1
for t in minw1 minw8; do for f in bench_$t bench_amazon_$t; do echo "== $f"; python scripts/brief.py < gpurun_out/r2g/$f.json; done; done
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" gpurun_out/r2g/parity_minw8.txt | tail -n 3
