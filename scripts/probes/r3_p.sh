cd $GRAFT_REPO_ROOT
python -m pytest tests/test_sharded.py tests/test_replicated.py tests/test_bench_contract.py -x -q -m gpu 2>&1 | grep -a -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -40
