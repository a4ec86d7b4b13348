#!/bin/bash
mkdir -p gpurun_out/r2e
run() { # tag
  timeout 300 python bench.py --no-cpu-baseline --hbm-leg on --epoch-leg off > gpurun_out/r2e/bench_$1.json 2> gpurun_out/r2e/bench_$1.err
  timeout 300 python bench.py --no-cpu-baseline --workload amazon-book --hbm-leg off --epoch-leg off > gpurun_out/r2e/bench_amazon_$1.json 2> gpurun_out/r2e/bench_amazon_$1.err
  VARS=5 CAPS=384,512,768,1024 BANDS=8 timeout 300 python scripts/spmm_bench.py yelp2018 64 > gpurun_out/r2e/caps_$1.txt 2>&1
}
run tail1
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r2e/parity_tail1.txt 2>&1; echo "rc=$?" >> gpurun_out/r2e/parity_tail1.txt
IDG_BUILD_DEFS="-DIDG_WALK_TAIL=0" python id-grec_amd/build.py --force > gpurun_out/r2e/build0.log 2>&1
run tail0
python id-grec_amd/build.py --force > gpurun_out/r2e/build1.log 2>&1
for t in tail1 tail0; do for f in bench_$t bench_amazon_$t; do echo "== $f"; python scripts/brief.py < gpurun_out/r2e/$f.json; done; cat gpurun_out/r2e/caps_$t.txt | grep -v amdgpu; done
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" gpurun_out/r2e/parity_tail1.txt | tail -n 5
