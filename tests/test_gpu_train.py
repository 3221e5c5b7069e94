"""GPU parity of the training path (backward kernels, trainer) against torch autograd through the fp32 oracle restatement.
Tolerances: bf16 operands / fp32 accumulation, relative L2 of each gradient tensor."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


@pytest.mark.parametrize("d,heads,n,batch", [(72, 4, 640, 2), (32, 4, 256, 1), (64, 2, 384, 2), (128, 1, 256, 1)])
def test_attention_backward(d, heads, n, batch):
    """dq / dk / dv of softmax(q k^T / sqrt d) v for an upstream gradient d_o, vs autograd in fp32 on the bf16-rounded operands"""
    from dfot_amd import capi
    g = torch.Generator().manual_seed(d + n)
    q, k, v = (torch.randn(batch, heads, n, d, generator=g) for _ in range(3))
    do = torch.randn(batch, n, heads * d, generator=g)
    ds = 64 if d <= 64 else 128
    scale = math.log2(math.e) / math.sqrt(d)
    bf = lambda t: t.to(torch.bfloat16)

    def pad(t):
        out = torch.zeros(batch, heads, n, ds, dtype=torch.bfloat16, device="cuda")
        out[..., :d] = t.cuda()
        return out
    qs = bf(q * scale)  # what the forward consumes; the oracle uses exactly this q (unscaled back in fp32)
    qd, kd, vd = pad(qs), pad(bf(k)), pad(bf(v))
    dod = bf(do).cuda().contiguous()
    o = torch.empty(batch, n, heads * d, dtype=torch.bfloat16, device="cuda")
    dq, dk, dv = (torch.full((batch, heads, n, ds), float("nan"), dtype=torch.bfloat16, device="cuda") for _ in range(3))
    capi.check(capi.lib.dfot_op_attention_bwd(capi.ptr(qd), capi.ptr(kd), capi.ptr(vd), capi.ptr(dod), capi.ptr(o), heads * d,
                                              capi.ptr(dq), capi.ptr(dk), capi.ptr(dv), batch, heads, n, d, capi.stream_ptr()))
    torch.cuda.synchronize()
    qr = (qs.float() / scale).requires_grad_()
    kr, vr = bf(k).float().requires_grad_(), bf(v).float().requires_grad_()
    ref = torch.softmax(qr @ kr.transpose(-1, -2) / math.sqrt(d), -1) @ vr
    ref = ref.transpose(1, 2).reshape(batch, n, heads * d)
    ref.backward(bf(do).float())
    assert rel(o.float().cpu(), ref.detach()) < 1.5e-2
    for name, got, want in (("dq", dq, qr.grad), ("dk", dk, kr.grad), ("dv", dv, vr.grad)):
        got = got.float().cpu()
        assert torch.isfinite(got).all(), name
        assert (got[..., d:] == 0).all(), name  # pad columns stay zero
        r = rel(got[..., :d], want)
        print(f"attention_bwd d={d} {name}: rel {r:.2e}")
        assert r < 2e-2, (name, r)
