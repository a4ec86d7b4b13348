# round 4, session P: the row split threshold (rows above it are cut into 64-entry segments combined in the published order)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4p
for i in 1 2; do
for t in 128 64 96 48; do
  python bench.py --split $t --scale-point off --hbm-leg off --epoch-leg off --no-cpu-baseline > gpurun_out/r4p/bench_split${t}_$i.json 2> /dev/null
  python scripts/brief.py split$t-$i < gpurun_out/r4p/bench_split${t}_$i.json
done
done
