#!/bin/bash
echo "== new"; python scripts/probes/infonce_ab.py 2>&1 | grep " us"
echo "== base (HEAD)"; IDG_LIB_PATH=$PWD/id-grec_amd/lib_base/libidgrec.so python scripts/probes/infonce_ab.py 2>&1 | grep " us"
python -m pytest tests/test_gpu_parity.py tests/test_gpu_models.py -x -q -k "infonce or ssl or simgcl or sgl or egcf" 2>&1 | tail -3
