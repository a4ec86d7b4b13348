# round 4, session B: the sharded step's new forms on the HIP kernels (K > 3, device-built id list, timeline), the
# watchdog's retry, the 2-rank rehearsal with its timeline, world-1 timings
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
timeout 1800 python -m pytest tests/test_sharded.py tests/test_bench_contract.py -x -q -m gpu > gpurun_out/r4b/pytest.txt 2>&1; echo "pytest rc=$?"
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" gpurun_out/r4b/pytest.txt | tail -15
WORKLOAD=synth-1M DIM=256 bash scripts/rehearse_shard2.sh --scale-point off
tail -5 gpurun_out/reh/shard2.err
for i in 1 2; do bash scripts/sharded1.sh --steps 300 --warmup 30 > gpurun_out/r4b/shard1_yelp_$i.json 2> gpurun_out/r4b/shard1_yelp_$i.err; python scripts/brief.py sh$i < gpurun_out/r4b/shard1_yelp_$i.json; done
