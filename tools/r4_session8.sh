#!/bin/bash
set -o pipefail
OUT=gpurun_out
python -m pytest tests -m gpu -q -x > $OUT/r4k_tests.log 2>&1 || { tail -40 $OUT/r4k_tests.log; exit 1; }
tail -3 $OUT/r4k_tests.log
bash tools/r4_profiles.sh
