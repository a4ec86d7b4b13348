#!/bin/bash
# round 2, first GPU session: the new at-size / any-width / k>64 tests, then the whole suite, then the bench
mkdir -p gpurun_out/r2a
timeout 900 python -m pytest tests/test_gpu_scale.py -x -q -m gpu > gpurun_out/r2a/scale.txt 2>&1; echo "rc=$?" >> gpurun_out/r2a/scale.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "topk_beyond or any_width" > gpurun_out/r2a/new_parity.txt 2>&1; echo "rc=$?" >> gpurun_out/r2a/new_parity.txt
timeout 600 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "without_tiled" > gpurun_out/r2a/d48.txt 2>&1; echo "rc=$?" >> gpurun_out/r2a/d48.txt
timeout 300 python bench.py > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err; echo "bench rc=$?" >> gpurun_out/r2a/bench.err
timeout 2400 python -m pytest tests -q -m gpu --deselect tests/test_gpu_scale.py > gpurun_out/r2a/pytest_all.txt 2>&1; echo "rc=$?" >> gpurun_out/r2a/pytest_all.txt
tail -5 gpurun_out/r2a/scale.txt gpurun_out/r2a/new_parity.txt gpurun_out/r2a/d48.txt gpurun_out/r2a/pytest_all.txt; cat gpurun_out/r2a/bench.json; tail -3 gpurun_out/r2a/bench.err
