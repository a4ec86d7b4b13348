# round 4, session C: fused EGCF step (EPI_ACT epilogues, cross InfoNCE) vs the reference goldens, epoch time, headline check
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "egcf" > gpurun_out/r4c/pytest_egcf.txt 2>&1; echo "pytest rc=$?"
tail -25 gpurun_out/r4c/pytest_egcf.txt
python scripts/e2e_epoch.py EGCF 4 2>&1 | grep -a "Training time\|Error\|error" | tail -5
python bench.py --scale-point off --hbm-leg off --epoch-leg off --no-cpu-baseline > gpurun_out/r4c/bench.json 2> gpurun_out/r4c/bench.err; python scripts/brief.py r4c < gpurun_out/r4c/bench.json
