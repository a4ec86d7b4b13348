# round 4, session E: graph build time after the multi-threaded passes; the whole GPU suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4e
python scripts/probes/graph_build_time.py synth-10M 2>&1 | grep -v "amdgpu.ids\|Warn\|torch.sparse_csr"
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r4e/pytest.txt 2>&1; echo "pytest rc=$?"
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" gpurun_out/r4e/pytest.txt | tail -15
