# round 4, session F: NGCF fused step parity + epoch time; compacted-input first backward product: parity + A/B of the step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f
timeout 1200 python -m pytest tests/test_gpu_models.py tests/test_gpu_parity.py -x -q -m gpu -k "ngcf or masked_backward or fused_step or trajectory or epilogue or egcf" > gpurun_out/r4f/pytest.txt 2>&1; echo "pytest rc=$?"
tail -25 gpurun_out/r4f/pytest.txt
python scripts/e2e_epoch.py NGCF 4 2>&1 | grep -a "Training time\|Error\|error" | tail -3
for i in 1 2 3; do
  for v in 0 1; do
    IDG_COMPACT_INPUTS=$v python bench.py --scale-point off --hbm-leg off --epoch-leg off --no-cpu-baseline > gpurun_out/r4f/bench_c${v}_$i.json 2> gpurun_out/r4f/bench_c${v}_$i.err
    python scripts/brief.py compact$v-$i < gpurun_out/r4f/bench_c${v}_$i.json
  done
done
