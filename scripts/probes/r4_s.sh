cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_models.py tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q -m gpu -k "egcf or sgl or simgcl or infonce or ssl" 2>&1 | tail -4
for e in EGCF SGL SimGCL XSimGCL; do python scripts/e2e_epoch.py $e 4 2>&1 | grep -a "Training time" | tail -1 | sed "s/^/$e /"; IDG_SSL_MFMA=0 python scripts/e2e_epoch.py $e 4 2>&1 | grep -a "Training time" | tail -1 | sed "s/^/$e simt /"; done
