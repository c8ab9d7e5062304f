#!/bin/bash
set -o pipefail
OUT=gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "w4_kernel" > $OUT/r4v_tests.log 2>&1; rc=$?
tail -25 $OUT/r4v_tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 120 python tools/gemm_ab.py 65536 base gemm_w4=0 2>&1 | tee $OUT/r4v_gemm_ab.log
