"""HBM-side traffic per launch of every hot kernel IN SITU: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate passes, counters
only) over one eager two-step C2 sample() (tools/sample_one.py), averaged per kernel name.  FETCH_SIZE reads 1/2 on gfx950 (MI355X_MICROARCH.md).
   python tools/pmc_insitu_traffic.py collect <outdir> [bench args...]     python tools/pmc_insitu_traffic.py summarise <outdir> [qkv_traffic.json]
With a json path, the in-situ figure of the fused QKV projection (round 4: gemm_w4_kernel<4, true>, the one-wave-per-SIMD LayerNorm-fold build; before it gemm_fast_kernel<256, 128, 0, 4, 30, 5, true>) is
written in the form bench.py reads for roofline.traffic."""
import collections
import csv
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def collect(outdir, extra):
    os.makedirs(outdir, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    for name, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        cmd = ["rocprofv3", "--pmc", ctr, "--output-format", "csv", "-d", os.path.join(outdir, name), "--", "python3", os.path.join(ROOT, "tools", "sample_one.py"), *extra]
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True)
        print(name, "rc", r.returncode, flush=True)
        if r.returncode != 0:
            print(r.stdout[-2000:], r.stderr[-2000:])
            sys.exit(1)


def summarise(outdir, json_out=None):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    rows = []
    for k, d in acc.items():
        if "FETCH_SIZE" not in d or "WRITE_SIZE" not in d:
            continue
        n = len(d["FETCH_SIZE"])
        fetch = sum(d["FETCH_SIZE"]) / n * 1024 * 2
        write = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"]) * 1024
        rows.append((fetch + write, k, n, fetch, write))
    rows.sort(reverse=True)
    print(f"{'kernel':<70s} {'launches':>8s} {'fetch MB':>10s} {'write MB':>10s} {'total MB':>10s}   (per launch, HBM side of L2)")
    for tot, k, n, fetch, write in rows[:18]:
        print(f"{k[:70]:<70s} {n:8d} {fetch / 1e6:10.1f} {write / 1e6:10.1f} {tot / 1e6:10.1f}")
    if json_out:
        import json
        qkv = ([r for r in rows if "gemm_w4_kernel<4, true" in r[1]] or [r for r in rows if "gemm_w4_kernel<4" in r[1]] or
               [r for r in rows if "gemm_fast_kernel<256, 128, 0, 4, 30, 5, true>" in r[1]] or [r for r in rows if "gemm_fast_kernel<256, 128, 0, 4, 30, 5" in r[1]])
        if not qkv:
            sys.exit("no fused-QKV launch of the persistent kernel in the counter files")
        tot, k, n, fetch, write = qkv[0]
        rec = {"qkv": {"rows": 65536, "seq_len": 1024, "bytes_per_launch": int(tot), "fetch_bytes": int(fetch), "write_bytes": int(write), "launches": n,
                       "kernel": k, "algorithmic_bytes": 65536 * 1024 * 2 + 3072 * 1024 * 2 + 65536 * 3072 * 2 + 65536 * 8,
                       "how": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (two separate passes) over one eager two-step C2 sample() (tools/sample_one.py), mean "
                              "over the launches of the fused QKV + RoPE projection IN SITU; FETCH_SIZE x 2 (gfx950)"}}
        with open(json_out, "w") as f:
            json.dump(rec, f, indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "collect":
        collect(sys.argv[2], sys.argv[3:])
    else:
        summarise(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
