#!/usr/bin/env bash
# rocprofv3 PMC passes on the attention kernels (separate passes, counters only -- no trace domains beside them).
# usage: tools/pmc_attn.sh <tag> <variant> ; results under gpurun_out/<tag>_pmc*/
set -u
tag="$1"; v="$2"
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INST_LEVEL_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_IFETCH SQ_IFETCH_LEVEL SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS_F32"; do
    i=$((i+1))
    timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d gpurun_out/${tag}_pmc$i -- python3 tools/attn_one.py $v 64 1024 16 3 > gpurun_out/${tag}_pmc$i.log 2>&1 || { echo "pass $i failed"; tail -5 gpurun_out/${tag}_pmc$i.log; }
done
python3 - <<PY
import csv, glob, collections
for i in (1,2,3,4):
    for f in glob.glob("gpurun_out/${tag}_pmc%d/**/*counter_collection.csv" % i, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "attn" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        for k, d in acc.items():
            print(k[:60]); 
            for c, val in sorted(d.items()): print(f"   {c:32s} {val:.4g}")
PY
