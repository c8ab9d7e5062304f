#!/bin/bash
set -o pipefail
timeout -k 10 300 tools/bin/gemm_w4_probe 2>&1 | tee gpurun_out/r4s_w4_probe.log
timeout -k 10 120 python tools/hipblaslt_ref.py 65536 2>&1 | tee gpurun_out/r4s_hipblaslt.log
python - <<'PY'
import torch
a=torch.randn(8192,8192,device="cuda",dtype=torch.bfloat16); b=torch.randn(8192,8192,device="cuda",dtype=torch.bfloat16)
for _ in range(3): torch.matmul(a,b.t())
torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): torch.matmul(a,b.t())
e1.record(); torch.cuda.synchronize()
ms=e0.elapsed_time(e1)/10
print(f"cube 8192^3 torch.matmul {ms*1e3:.1f} us = {2*8192**3/ms/1e9:.0f} TF")
PY
