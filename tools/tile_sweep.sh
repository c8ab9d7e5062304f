#!/usr/bin/env bash
# which tile wins at which token count (isolated call sites, warm operands): tools/gemm_ab.py with the tile forced
set -u
cd "$(dirname "$0")/.."
for rows in ${@:-2048 3072 4096 6144 8192 12288 16384 24576}; do
  timeout -k 10 120 python tools/gemm_ab.py $rows base gemm_tile=256256 gemm_tile=256128 gemm_tile=256064 gemm_tile=128128 gemm_tile=128064 2>&1 | tail -4
done
