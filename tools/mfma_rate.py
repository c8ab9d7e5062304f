#!/usr/bin/env python3
"""Sustained v_mfma_f32_16x16x32_bf16 rate of the device (f5_bench_mfma_rate), repeated to see the power-management settle."""
import ctypes as C, sys, time
import torch
sys.path.insert(0, ".")
from eraxvif5tts_amd import _lib
_lib.require_gpu()
lib = _lib.load()
for rnd in (0, 1, 1, 1, 1, 1, 1, 1, 1, 0):
    tf = C.c_float()
    t0 = time.perf_counter()
    _lib.check(lib.f5_bench_mfma_rate(rnd, C.byref(tf), _lib.stream_ptr()))
    print(f"{'random' if rnd else 'zero  '} operands: {tf.value:7.1f} TFLOP/s  ({(time.perf_counter() - t0) * 1e3:.0f} ms per call)", flush=True)
