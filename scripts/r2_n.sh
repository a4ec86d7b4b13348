#!/bin/bash
mkdir -p gpurun_out/r2n
run() {
  timeout 300 python bench.py --no-cpu-baseline --hbm-leg off --epoch-leg off > gpurun_out/r2n/bench_$1.json 2> gpurun_out/r2n/bench_$1.err
  timeout 300 python bench.py --no-cpu-baseline --workload amazon-book --hbm-leg off --epoch-leg off > gpurun_out/r2n/bench_amazon_$1.json 2> gpurun_out/r2n/bench_amazon_$1.err
  timeout 300 python bench.py --no-cpu-baseline --model MFBPR --batch 2048 --epoch-leg off > gpurun_out/r2n/bench_mfbpr_$1.json 2> gpurun_out/r2n/bench_mfbpr_$1.err
}
run u8
for u in 16 32; do
  IDG_BUILD_DEFS="-DIDG_UNITS_UNROLL=$u" python id-grec_amd/build.py --force > gpurun_out/r2n/build_u$u.log 2>&1
  run u$u
done
python id-grec_amd/build.py --force > gpurun_out/r2n/build_default.log 2>&1
for t in u8 u16 u32; do for f in bench_$t bench_amazon_$t bench_mfbpr_$t; do echo "== $f"; python scripts/brief.py < gpurun_out/r2n/$f.json; done; done
