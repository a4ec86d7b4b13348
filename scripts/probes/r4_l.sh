cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4l
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "graph or spmm or propagate" > gpurun_out/r4l/pytest.txt 2>&1; echo "pytest rc=$?"
tail -8 gpurun_out/r4l/pytest.txt
python scripts/probes/graph_build_time.py synth-10M 2>&1 | grep -v "amdgpu.ids\|Warn\|torch.sparse_csr"
