cd $GRAFT_REPO_ROOT
for n in 1 2 3 4 6; do
IDG_BUILD_DEFS="-DIDG_TOPK_FLOOR_SLABS=$n" python id-grec_amd/build.py --force > /dev/null 2>&1
echo "floor slabs $n: $(python scripts/eval_bench.py yelp2018 2>&1 | grep 'idg_score_topk_f32')"
echo "floor slabs $n: $(python scripts/eval_bench.py amazon-book 2>&1 | grep 'idg_score_topk_f32')"
done
