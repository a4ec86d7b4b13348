#!/bin/bash
python -m pytest tests/test_gpu_models.py -x -q -k "lookahead" 2>&1 | tail -3
bash scripts/prof.sh r04_simgcl_amazon --workload amazon-book --model SimGCL --batch 2048 --scale-point off > /dev/null 2>&1
grep -a ms_per_step gpurun_out/prof_r04_simgcl_amazon.log | tail -1 | cut -c1-160
