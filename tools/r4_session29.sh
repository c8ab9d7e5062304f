#!/bin/bash
set -o pipefail
OUT=gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=6 > $OUT/r4ap_tests.log 2>&1 || { tail -40 $OUT/r4ap_tests.log; exit 1; }
tail -10 $OUT/r4ap_tests.log
