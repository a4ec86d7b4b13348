# end-of-round profiles: kernel statistics of the default line and of the sharded step at world size 1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R && python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "device_tensor" 2>&1 | tail -2; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03_yelp -o t -- python3 $R/bench.py --no-cpu-baseline --hbm-leg off --epoch-leg off --scale-point off > $R/gpurun_out/prof_r03_yelp.log 2>&1
rm -f $R/gpurun_out/prof_r03_yelp/*kernel_trace.csv
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29512
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03_shard1_yelp -o t -- python3 $R/bench.py --gpus 1 --force-sharded --parallel shard --no-cpu-baseline --workload yelp2018 --steps 300 --warmup 30 --ramp gemm > $R/gpurun_out/prof_r03_shard1_yelp.log 2>&1
rm -f $R/gpurun_out/prof_r03_shard1_yelp/*kernel_trace.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03_shard1_c5 -o t -- python3 $R/bench.py --gpus 1 --force-sharded --parallel shard --no-cpu-baseline --workload synth-10M --dim 256 --steps 6 --warmup 3 --ramp gemm > $R/gpurun_out/prof_r03_shard1_c5.log 2>&1
rm -f $R/gpurun_out/prof_r03_shard1_c5/*kernel_trace.csv
grep -h '^{' $R/gpurun_out/prof_r03_shard1_c5.log | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('c5 shard1 under tracer', d['ms_per_step'])"
