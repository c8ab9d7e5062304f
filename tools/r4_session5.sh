#!/bin/bash
# round 4, GPU session 5: the whole GPU suite (durations), then the one confirming rocprofv3 --pmc run of bench.py
set -o pipefail
OUT=gpurun_out
python -m pytest tests -m gpu -q --durations=25 -x > $OUT/r4f_tests.log 2>&1 || { tail -60 $OUT/r4f_tests.log; exit 1; }
tail -32 $OUT/r4f_tests.log
ROOTD=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $ROOTD/$OUT/r4f_pmc -- python3 $ROOTD/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-graph --no-extra > $ROOTD/$OUT/r4f_pmc_bench.log 2>&1
echo "rocprofv3 --pmc bench.py exit code $?"
tail -c 1200 $ROOTD/$OUT/r4f_pmc_bench.log
ls $ROOTD/$OUT/r4f_pmc | head; du -sh $ROOTD/$OUT/r4f_pmc
find $ROOTD/$OUT/r4f_pmc -name "*.csv" -size +4M -delete; find $ROOTD/$OUT/r4f_pmc -name "*.db" -delete
