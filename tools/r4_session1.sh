#!/bin/bash
# round 4, GPU session 1: op/fullsize/sharded tests on the new GEMM launch split, M % 256 != 0 bench rows (A/B of the split), default bench line,
# then the one confirming rocprofv3 --pmc run of bench.py -- last, because it is the step that used to die
set -o pipefail
OUT=gpurun_out
python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_sharded.py -m gpu -q -x > $OUT/r4b_tests.log 2>&1 || { tail -30 $OUT/r4b_tests.log; exit 1; }
tail -3 $OUT/r4b_tests.log
python -m pytest tests/test_gpu_vocoder_wrapper.py tests/test_gpu_model.py -m gpu -q -x -s -k "oracle_chain or deferred_range_guard" > $OUT/r4b_tests2.log 2>&1 || { tail -30 $OUT/r4b_tests2.log; exit 1; }
grep -a "rel-L2" $OUT/r4b_tests2.log; tail -2 $OUT/r4b_tests2.log
for cfgv in "1024 1" "1001 1" "1001 0"; do
  set -- $cfgv
  F5HIP_TUNING=gemm_split_tail=$2 python bench.py --batch 8 --seq-len $1 --no-extra --no-cpu-baseline --steps 3 --warmup 1 > $OUT/r4b_bench_8x$1_split$2.json 2> $OUT/r4b_bench_8x$1_split$2.err || exit 1
  python - <<PY
import json
j=json.loads(open("$OUT/r4b_bench_8x$1_split$2.json").read().strip().splitlines()[-1])
print("8x$1 split=$2", j["value"], "mel-frames/s", {k["kernel"]: k["ms"] for k in j["roofline"]["kernels"]})
PY
done
python bench.py > $OUT/r4b_bench_default.json 2> $OUT/r4b_bench_default.err || exit 1
python - <<PY
import json
j=json.loads(open("$OUT/r4b_bench_default.json").read().strip().splitlines()[-1])
print("C2", j["value"], j["ms_per_step"], {k["kernel"]: k["ms"] for k in j["roofline"]["kernels"]})
print({k: (v.get("value"), v.get("ms_per_step")) for k, v in j["workloads"].items()})
PY
ROOTD=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $ROOTD/$OUT/r4b_pmc -- python3 $ROOTD/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-graph --no-extra > $ROOTD/$OUT/r4b_pmc_bench.log 2>&1
echo "rocprofv3 --pmc bench.py exit code $?"
tail -c 1500 $ROOTD/$OUT/r4b_pmc_bench.log
ls $ROOTD/$OUT/r4b_pmc | head; du -sh $ROOTD/$OUT/r4b_pmc
find $ROOTD/$OUT/r4b_pmc -name "*.csv" -size +4M -delete; find $ROOTD/$OUT/r4b_pmc -name "*.db" -delete
