"""Times dfot_op_attention_bwd_lse (delta + dQ + dK/dV kernels) at a training launch shape with HIP events on the launch stream.
usage (GPU box): python tools/bench_attn_bwd.py [batch heads n d] ; A/B through the library's DFOT_* switches in separate processes"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dfot_amd  # noqa: E402,F401
from dfot_amd import capi  # noqa: E402


def main():
    b, h, n, d = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (8, 9, 8192, 64)
    g = torch.Generator(device="cuda").manual_seed(3)
    nrm = lambda t: t / t.pow(2).mean(-1, keepdim=True).sqrt()
    scale = math.log2(math.e) / math.sqrt(d)
    q = (nrm(torch.randn(b, h, n, d, device="cuda", generator=g)) * scale).to(torch.bfloat16)
    k = nrm(torch.randn(b, h, n, d, device="cuda", generator=g)).to(torch.bfloat16)
    v = torch.randn(b, h, n, d, device="cuda", generator=g).to(torch.bfloat16)
    do = torch.randn(b, n, h * d, device="cuda", generator=g).to(torch.bfloat16)
    o = torch.empty(b, n, h * d, dtype=torch.bfloat16, device="cuda")
    lse = torch.empty(b, h, n, dtype=torch.float32, device="cuda")
    need = int(capi.lib.dfot_op_attention_scratch_bytes(b, h, n, d))
    scratch = torch.empty(max(need, 1), dtype=torch.uint8, device="cuda")
    P, S = capi.ptr, capi.stream_ptr
    capi.check(capi.lib.dfot_op_attention_fwd_lse_bounded(P(q), P(k), P(v), P(o), h * d, P(lse), b, h, n, d, 20.0, P(scratch) if need else None, need, S()))
    delta = torch.empty_like(lse)
    dq, dk, dv = (torch.empty_like(q) for _ in range(3))

    def run():
        capi.check(capi.lib.dfot_op_attention_bwd_lse(P(q), P(k), P(v), P(o), P(do), h * d, P(lse), P(delta), P(dq), P(dk), P(dv), b, h, n, d, S()))
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    reps = 6
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    fl = 10.0 * b * h * n * n * d
    print(f"attention_bwd ({b},{h},{n},{d}) {os.environ.get('DFOT_ATTN_BWD_PIPE', 'default')}: {us:.1f} us  {fl / us / 1e6:.0f} TF/s algorithmic; "
          f"checksum dq {float(dq.float().abs().sum()):.6e} dk {float(dk.float().abs().sum()):.6e} dv {float(dv.float().abs().sum()):.6e}")


if __name__ == "__main__":
    main()
