cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4n
timeout 1200 python -m pytest tests/test_gpu_models.py tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q -m gpu -k "egcf or sgl or simgcl or infonce or ssl" > gpurun_out/r4n/pytest.txt 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r4n/pytest.txt
for e in EGCF SGL SimGCL; do python scripts/e2e_epoch.py $e 4 2>&1 | grep -a "Training time" | tail -1 | sed "s/^/$e /"; done
