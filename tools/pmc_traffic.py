"""HBM-side traffic of the dominant GEMM from rocprofv3 PMC passes (separate passes, as MI355X_MICROARCH.md prescribes; FETCH_SIZE is doubled on
gfx950, WRITE_SIZE is exact for 16-byte stores).  Run on the GPU box:
    python tools/pmc_traffic.py collect gpurun_out/r2_traffic     (two rocprofv3 --pmc runs of tools/gemm_site.py 0)
    python tools/pmc_traffic.py summarise gpurun_out/r2_traffic profiles/r2_qkv_traffic.json
bench.py reads the newest profiles/*traffic*.json for roofline.traffic."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys


def collect(outdir):
    env = dict(os.environ, TMPDIR="/tmp")
    for name, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE"), ("tcc", "TCC_HIT_sum TCC_MISS_sum")):
        cmd = ["rocprofv3", "--pmc", *ctr.split(), "--output-format", "csv", "-d", os.path.join(outdir, name), "--", sys.executable, "tools/gemm_site.py", "0"]
        print(" ".join(cmd), flush=True)
        subprocess.run(cmd, env=env, check=False, timeout=300)


def summarise(outdir, dst):
    vals = collections.defaultdict(list)
    for f in glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_fast_kernel" in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
        sys.exit(f"no FETCH_SIZE / WRITE_SIZE rows for gemm_fast_kernel under {outdir}")
    mean = {k: sum(v) / len(v) for k, v in vals.items()}
    fetch_b, write_b = mean["FETCH_SIZE"] * 1024 * 2, mean["WRITE_SIZE"] * 1024  # counters are in KiB; FETCH_SIZE reads 1/2 on gfx950
    rec = {"qkv": {"rows": 65536, "seq_len": 1024, "bytes_per_launch": int(fetch_b + write_b), "fetch_bytes": int(fetch_b), "write_bytes": int(write_b),
                   "launches": len(vals["FETCH_SIZE"]),
                   "l2_hit_rate": (mean["TCC_HIT_sum"] / (mean["TCC_HIT_sum"] + mean["TCC_MISS_sum"])) if "TCC_HIT_sum" in mean else None,
                   "algorithmic_bytes": 65536 * 1024 * 2 + 3072 * 1024 * 2 + 65536 * 3072 * 2,
                   "how": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum (three separate passes) on tools/gemm_site.py 0 "
                          "(f5_bench_gemm_site: fused QKV + RoPE, M=65536 N=3072 K=1024, random bf16 operands); FETCH_SIZE x 2 (gfx950)"}}
    with open(dst, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "collect":
        collect(sys.argv[2])
    else:
        summarise(sys.argv[2], sys.argv[3])
