# round 4, session A: the default line with the synth-10M d=64 HBM-bound leg; oracle pin at configs[4]'s real shape;
# kernel stats + PMC passes of the synth-10M d=64 dense launch (profiles/r04)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
python bench.py > gpurun_out/r4a/bench_default.json 2> gpurun_out/r4a/bench_default.err; echo "bench rc=$?"
python scripts/brief.py r4a < gpurun_out/r4a/bench_default.json | tail -3
timeout 1500 python -m pytest tests/test_gpu_scale.py -x -q -m gpu -k "config5_shape" > gpurun_out/r4a/pytest_c5.txt 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r4a/pytest_c5.txt
bash scripts/prof.sh r04_synth10M_d64 --workload synth-10M --dim 64 --steps 6 --warmup 3 --ramp gemm --scale-point off --separate-adam
cd $GRAFT_REPO_ROOT
ls gpurun_out/prof_r04_synth10M_d64/ | head
bash scripts/pmc.sh r04_synth10M_d64 --workload synth-10M --dim 64 --separate-adam --scale-point off > gpurun_out/r4a/pmc.txt 2>&1
cd $GRAFT_REPO_ROOT
tail -30 gpurun_out/r4a/pmc.txt
