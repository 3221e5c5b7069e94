"""Per-op timing table of one RE10K training step (BASELINE config 5): every C-ABI call of the step is bracketed by two events on the
launch stream and summed by (entry point, integer arguments = its shape).  Diagnostic, not a bench: the brackets serialise nothing (one
stream) but add ~5 us of host work per call.    usage (GPU box): python tools/train_op_table.py [batch] > gpurun_out/train_ops.txt"""
import collections
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (registers the dfot_amd alias)
import dfot_amd  # noqa: E402
from dfot_amd import capi, uvit_train  # noqa: E402


class TimedLib:
    def __init__(self, lib):
        self._lib, self.records, self.on = lib, [], False

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if not name.startswith("dfot_op_") or name.endswith("_bytes"):
            return fn

        def call(*a):
            if not self.on:
                return fn(*a)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a)
            e1.record()
            ints = tuple(x for x in a if isinstance(x, int) and not isinstance(x, bool) and x < (1 << 40))
            self.records.append((name, ints, e0, e1))
            return r
        return call


def main():
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    timed = TimedLib(capi.lib)
    capi.lib = timed
    init = dfot_amd.UViT3DPose(bench.RE10K, x_shape=(3, 256, 256), max_tokens=8)
    init.init_random(seed=0)
    cfg = dict(bench.RE10K, resolution=256, max_tokens=8, in_channels=3, cond_dim=180, noise_dim=256)
    tr = uvit_train.UViT3DPoseTrainer({k: v.detach() for k, v in init.state_dict().items()}, cfg)
    del init
    g = torch.Generator().manual_seed(200)
    xs, noise = torch.randn(b, 8, 3, 256, 256, generator=g).cuda(), torch.randn(b, 8, 3, 256, 256, generator=g).cuda()
    cond = torch.ops.dfot.ray_encoding(bench.synth_poses(b, 8, 300), 256)
    tn = dfot_amd.TrainingNoise(noise_level="random_independent", is_continuous=True, n_context_tokens=1)
    masks = torch.ones(b, 8, dtype=torch.bool)
    for i in range(3):
        lv = tn.sample(b, 8, masks, g, training=True)
        timed.on = i == 2
        torch.cuda.synchronize()
        tr.loss_and_grads(xs, cond, lv[0], noise, lv[1])
        tr.optimizer_step(lr=5e-5, betas=(0.9, 0.99), weight_decay=0.01, max_grad_norm=1.0)
    torch.cuda.synchronize()
    agg = collections.OrderedDict()
    for name, ints, e0, e1 in timed.records:
        k = (name, ints)
        t = e0.elapsed_time(e1) * 1e3
        c = agg.setdefault(k, [0, 0.0])
        c[0] += 1
        c[1] += t
    tot = sum(v[1] for v in agg.values())
    print(f"# {len(timed.records)} calls, {tot / 1e3:.2f} ms inside op brackets")
    for (name, ints), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{name[8:]:28s} {str(ints):58s} x{n:<4d} avg {t / n:9.1f} us  total {t / 1e3:7.2f} ms  {100 * t / tot:5.1f}%")


if __name__ == "__main__":
    main()
