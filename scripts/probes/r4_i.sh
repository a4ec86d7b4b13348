# round 4, session I: top-K with the floor phase inside the main launch: parity, timing vs the two-launch form
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4i
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q -m gpu -k "topk" > gpurun_out/r4i/pytest.txt 2>&1; echo "pytest rc=$?"
tail -6 gpurun_out/r4i/pytest.txt
for f in 1 2 0; do echo "IDG_TOPK_FLOOR=$f"; IDG_TOPK_FLOOR=$f timeout 300 python scripts/eval_bench.py yelp2018 2>&1 | grep "idg_score\|one call\|rows with"; done
for f in 1 2; do echo "IDG_TOPK_FLOOR=$f amazon-book"; IDG_TOPK_FLOOR=$f timeout 300 python scripts/eval_bench.py amazon-book 2>&1 | grep "idg_score"; done
