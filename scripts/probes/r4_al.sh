#!/bin/bash
bash scripts/prof.sh r04_yelp2 --scale-point off --steps 1000 --warmup 100 > /dev/null 2>&1
grep -a "ms_per_step" gpurun_out/prof_r04_yelp2.log | tail -1 | cut -c1-200
