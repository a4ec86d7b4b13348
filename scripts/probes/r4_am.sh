#!/bin/bash
# PMC passes of the default (yelp2018-shape) and amazon-book-shape runs for profiles/r04/traffic_*.json + their kernel stats
bash scripts/pmc.sh r04_yelp --separate-adam --scale-point off > /dev/null 2>&1
bash scripts/prof.sh r04_yelp_sepadam --separate-adam --scale-point off > /dev/null 2>&1
bash scripts/pmc.sh r04_amazon --workload amazon-book --separate-adam --scale-point off > /dev/null 2>&1
bash scripts/prof.sh r04_amazon_sepadam --workload amazon-book --separate-adam --scale-point off > /dev/null 2>&1
ls gpurun_out/pmc_r04_yelp gpurun_out/pmc_r04_amazon | head -20
