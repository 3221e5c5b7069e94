#!/usr/bin/env python3
"""Micro-benchmarks of the C-ABI primitives at the RE10K model shapes (model batch 2), HIP-event timed.
Usage (GPU box): python tools/bench_ops.py [gemm] [conv] [attn]"""
import ctypes as C
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dfot_amd  # noqa: E402
from dfot_amd import capi  # noqa: E402

S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr())


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def gemm(m, n, k, variant):
    a = torch.randn(m, k, device="cuda").bfloat16()
    w = (torch.randn(n, k, device="cuda") / math.sqrt(k)).bfloat16()
    out = torch.empty(m, n, device="cuda")
    ms = timeit(lambda: capi.check(capi.lib.dfot_op_gemm(P(a), k, P(w), None, P(out), m, n, k, variant, S())))
    return ms, 2.0 * m * n * k / ms / 1e9


def conv(bt, h, w, cin, cout, variant):
    a = torch.randn(bt, h, w, cin, device="cuda").bfloat16()
    wt = (torch.randn(cout, 9 * cin, device="cuda") / math.sqrt(9 * cin)).bfloat16()
    out = torch.empty(bt, h, w, cout, device="cuda")
    ms = timeit(lambda: capi.check(capi.lib.dfot_op_conv3x3(P(a), P(wt), None, P(out), bt, h, w, cin, cout, variant, S())))
    return ms, 2.0 * bt * h * w * cout * 9 * cin / ms / 1e9


def attn(b, heads, n, d, variant):
    q = torch.randn(b, heads, n, d, device="cuda").bfloat16() * 0.2
    k = torch.randn(b, heads, n, d, device="cuda").bfloat16()
    v = torch.randn(b, heads, n, d, device="cuda").bfloat16()
    o = torch.empty(b, n, heads * d, device="cuda", dtype=torch.bfloat16)
    ms = timeit(lambda: capi.check(capi.lib.dfot_op_attention(P(q), P(k), P(v), P(o), heads * d, b, heads, n, d, variant, S())))
    return ms, 4.0 * b * heads * n * n * d / ms / 1e9


def main():
    what = sys.argv[1:] or ["gemm", "conv", "attn"]
    variants = [int(x) for x in os.environ.get("VARIANTS", "1").split(",")]
    if "gemm" in what:
        for name, (m, n, k) in {"L2 qkv+mlp": (16384, 4032, 576), "L2 out": (16384, 576, 2880), "L3 qkv+mlp": (4096, 8064, 1152),
                                "L3 out": (4096, 1152, 5760), "film L0": (131072, 256, 1024), "pose": (131072, 1024, 768),
                                "square 4096": (4096, 4096, 4096), "DiT qkv B8": (10240, 3456, 1152), "DiT proj B8": (10240, 1152, 1152),
                                "DiT qkv B2": (2560, 3456, 1152), "DiT proj B2": (2560, 1152, 1152),
                                "MLP fc1 B8": (20480, 4608, 1152), "MLP fc2 B8": (20480, 1152, 4608)}.items():
            if os.environ.get("ONLY") and os.environ["ONLY"] not in name:
                continue
            for v in variants:
                ms, tf = gemm(m, n, k, v)
                print(f"gemm {name:12s} M={m:6d} N={n:5d} K={k:5d} variant={v}: {ms*1e3:8.1f} us  {tf:7.1f} TF/s", flush=True)
            if os.environ.get("HIPBLASLT"):  # vendor library on the same shape, for orientation only (never on the product path)
                a = torch.randn(m, k, device="cuda").bfloat16()
                w = torch.randn(n, k, device="cuda").bfloat16()
                ms = timeit(lambda: torch.nn.functional.linear(a, w))
                print(f"gemm {name:12s} M={m:6d} N={n:5d} K={k:5d} torch/hipBLASLt: {ms*1e3:8.1f} us  {2.0*m*n*k/ms/1e9:7.1f} TF/s", flush=True)
    if "conv" in what:
        for name, (bt, h, w, ci, co) in {"L0 res": (16, 128, 128, 128, 128), "L1 res": (16, 64, 64, 256, 256),
                                         "down0": (16, 64, 64, 128, 256), "down1": (16, 32, 32, 256, 576),
                                         "down2": (16, 16, 16, 576, 1152), "up2": (16, 16, 16, 1152, 576),
                                         "up1": (16, 32, 32, 576, 256), "up0": (16, 64, 64, 256, 128)}.items():
            for v in variants:
                ms, tf = conv(bt, h, w, ci, co, v)
                print(f"conv {name:8s} {bt}x{h}x{w} {ci}->{co} variant={v}: {ms*1e3:8.1f} us  {tf:7.1f} TF/s", flush=True)
    if "attn" in what:
        shapes = {"L2": (2, 9, 8192, 64), "L3": (2, 9, 2048, 128), "L2 Bm8": (8, 9, 8192, 64)}
        if os.environ.get("ATTN_SHAPE"):  # e.g. ATTN_SHAPE=2,8,2048,128: one extra shape (workgroup-count experiments)
            shapes = {"custom": tuple(int(x) for x in os.environ["ATTN_SHAPE"].split(","))}
        for name, (b, hd, n, d) in shapes.items():
            if os.environ.get("ONLY") and os.environ["ONLY"] != name:
                continue
            if os.environ.get("ONLY") and os.environ["ONLY"] != name:
                continue
            for v in [int(x) for x in os.environ.get("ATTN_VARIANTS", "0").split(",")]:
                ms, tf = attn(b, hd, n, d, v)
                print(f"attn {name:7s} B={b} H={hd} N={n} d={d} variant={v}: {ms*1e3:8.1f} us  {tf:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
