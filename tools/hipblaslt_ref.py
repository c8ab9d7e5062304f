"""What the vendor library (hipBLASLt / rocBLAS behind torch.matmul) reaches at the four block GEMM shapes of C2, for comparison with
   tools/gemm_ab.py (plain GEMM, no fused epilogue; random bf16 operands; rotating buffers so weights are not L2-resident across calls):
   python tools/hipblaslt_ref.py [ROWS]"""
import sys
import torch

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda:0")
SITES = [("qkv", 3072, 1024), ("ff1", 2048, 1024), ("ff2", 1024, 2048), ("out", 1024, 1024)]
for name, N, K in SITES:
    nbuf = 4
    A = [torch.randn(rows, K, device=dev, dtype=torch.bfloat16) for _ in range(nbuf)]
    W = [torch.randn(N, K, device=dev, dtype=torch.bfloat16) for _ in range(nbuf)]
    out = torch.empty(rows, N, device=dev, dtype=torch.bfloat16)
    for i in range(6):
        torch.matmul(A[i % nbuf], W[i % nbuf].t(), out=out)
    torch.cuda.synchronize()
    best = 1e9
    for rnd in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(20):
            torch.matmul(A[i % nbuf], W[i % nbuf].t(), out=out)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    print(f"{name} M={rows} N={N} K={K}: torch.matmul {best * 1e3:.1f} us = {2.0 * rows * N * K / best / 1e9:.0f} TF", flush=True)
