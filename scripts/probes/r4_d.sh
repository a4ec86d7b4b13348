# round 4, session D: A/B of the library before / after the EPI_ACT epilogue on one box (headline step), NGCF epoch baseline
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4d
for i in 1 2 3; do
  for v in base new; do
    if [ $v = base ]; then export IDG_LIB_PATH=$GRAFT_REPO_ROOT/id-grec_amd/lib_base/libidgrec.so; else unset IDG_LIB_PATH; fi
    python bench.py --scale-point off --hbm-leg off --epoch-leg off --no-cpu-baseline > gpurun_out/r4d/bench_${v}_$i.json 2> gpurun_out/r4d/bench_${v}_$i.err
    python scripts/brief.py ${v}$i < gpurun_out/r4d/bench_${v}_$i.json
  done
done
unset IDG_LIB_PATH
python scripts/e2e_epoch.py NGCF 3 2>&1 | grep -a "Training time" | tail -2
