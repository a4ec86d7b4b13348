# round 4, session M: final artefacts — the default line, per-kernel stats of the default run and of the EGCF / NGCF epochs,
# the other BASELINE configs, the world-1 sharded lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4m
( time python bench.py --steps 20 --warmup 5 ) > gpurun_out/r4m/bench_default_20_5.json 2> gpurun_out/r4m/bench_default_20_5.err
python scripts/brief.py driver-form < gpurun_out/r4m/bench_default_20_5.json
tail -4 gpurun_out/r4m/bench_default_20_5.err
python bench.py > gpurun_out/r4m/bench_default.json 2> gpurun_out/r4m/bench_default.err
python scripts/brief.py default < gpurun_out/r4m/bench_default.json
bash scripts/prof.sh r04_yelp --scale-point off
cd $GRAFT_REPO_ROOT
for m in EGCF NGCF; do
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r04_$m -o $m -- python3 $GRAFT_REPO_ROOT/scripts/e2e_epoch.py $m 3 > $GRAFT_REPO_ROOT/gpurun_out/prof_r04_$m.log 2>&1
  cd $GRAFT_REPO_ROOT
  grep -a "Training time" gpurun_out/prof_r04_$m.log | tail -1
done
python bench.py --model MFBPR --batch 2048 --scale-point off --hbm-leg off > gpurun_out/r4m/bench_mfbpr.json 2>/dev/null; python scripts/brief.py mfbpr < gpurun_out/r4m/bench_mfbpr.json
python bench.py --workload amazon-book --scale-point off --hbm-leg off > gpurun_out/r4m/bench_amazon.json 2>/dev/null; python scripts/brief.py amazon < gpurun_out/r4m/bench_amazon.json
python bench.py --workload amazon-book --model SimGCL --batch 2048 --scale-point off --hbm-leg off > gpurun_out/r4m/bench_simgcl_amazon.json 2>/dev/null; python scripts/brief.py simgcl < gpurun_out/r4m/bench_simgcl_amazon.json
bash scripts/sharded1.sh --workload synth-10M --dim 256 --steps 6 --warmup 3 > gpurun_out/r4m/shard_world1_c5.json 2> gpurun_out/r4m/shard_world1_c5.err; python scripts/brief.py shard-c5 < gpurun_out/r4m/shard_world1_c5.json
for e in LightGCN SimGCL XSimGCL SGL MFBPR; do python scripts/e2e_epoch.py $e 3 2>&1 | grep -a "Training time" | tail -1 | sed "s/^/$e /"; done
