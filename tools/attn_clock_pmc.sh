#!/usr/bin/env bash
# Effective clock under the attention kernel from rocprofv3: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration, on LONG launches
# (B*2 = 256, N = 4096: ~17 ms each; the quotient reads high on short ones -- MI355X_MICROARCH.md, DVFS give-back).  Two passes: kernel trace, counter.
set -u
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
v="${1:-2}"
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/attnclk_trace -- python3 tools/attn_one.py $v 256 4096 16 6 > gpurun_out/attnclk_trace.log 2>&1 || tail -5 gpurun_out/attnclk_trace.log
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/attnclk_pmc -- python3 tools/attn_one.py $v 256 4096 16 6 > gpurun_out/attnclk_pmc.log 2>&1 || tail -5 gpurun_out/attnclk_pmc.log
python3 - <<'PY'
import csv, glob
dur = []
for f in glob.glob("gpurun_out/attnclk_trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attn" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
act = []
for f in glob.glob("gpurun_out/attnclk_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attn" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            act.append(float(r["Counter_Value"]))
if dur and act:
    d = sorted(dur)[len(dur) // 2]; a = sorted(act)[len(act) // 2]
    print(f"attention launches: {len(dur)} traced (median {d * 1e3:.3f} ms), {len(act)} counted (median GRBM_GUI_ACTIVE {a:.4g})")
    print(f"effective clock = GRBM_GUI_ACTIVE / 8 / duration = {a / 8 / d / 1e9:.3f} GHz   (counter pass and trace pass are separate runs of the same launches)")
    flops = 4.0 * 256 * 16 * 4096 * 4096 * 64
    print(f"{flops / d / 1e12:.0f} TFLOP/s; MFMA peak at that clock {1024 * 1024 * a / 8 / d / 1e12:.0f} TFLOP/s")
else:
    print("no data", len(dur), len(act))
PY
