#!/bin/bash
# in-situ sweep of the tile-walk order per call site under the LayerNorm fold (bit 0 qkv, 1 out, 2 ff1, 3 ff2): the fold changed who produces the
# operand each GEMM reads (QKV / FF1 now read the stream the residual GEMM in front of them just wrote)
OUT=gpurun_out
for rev in 8 9 12 13 0 10 14 11 8; do
  F5HIP_TUNING=gemm_reverse_sites=$rev python bench.py --no-extra --no-cpu-baseline --steps 3 --warmup 1 > $OUT/r4m_rev$rev.json 2> $OUT/r4m_rev$rev.err || { tail -5 $OUT/r4m_rev$rev.err; exit 1; }
  python - <<PY
import json
j=json.loads(open("$OUT/r4m_rev$rev.json").read().strip().splitlines()[-1])
k={x["kernel"][:8]: x["ms"] for x in j["roofline"]["kernels"]}
print("reverse_sites=$rev", j["value"], "qkv", k["qkv"], "attn", k["attentio"], "out", k["attn_out"], "ff1", k["ff1"], "ff2", k["ff2"])
PY
done
