#!/usr/bin/env python3
"""ML_MATH_F32X3 (fp32 tensors, split-operand products on the f16 matrix pipe) against ML_MATH_F32 on the same fp32
tensors: error of both against an fp64 convolution of the same operands (small shapes, CPU double), and launch times on
the shapes of the 8 x 1024^2 ResNeXt-50 forward.  GPU box: python scripts/x3_probe.py [--no-accuracy]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instance-segmentation-road-project_amd")]
import numpy as np
import torch
import torch.nn.functional as F

from masklab_hip import _lib, ops, packing

# (label, (B, H, W), cin, cout, k, stride, residual)
TIMED = [
    ("tower 3x3 256->256 P3", (8, 128, 128), 256, 256, 3, 1, False),
    ("tower 3x3 256->256 P4", (8, 64, 64), 256, 256, 3, 1, False),
    ("fpn 3x3 256->256 P3", (8, 128, 128), 256, 256, 3, 1, False),
    ("s1 conv1 256->128", (8, 256, 256), 256, 128, 1, 1, False),
    ("s1 conv3 128->256 +res", (8, 256, 256), 128, 256, 1, 1, True),
    ("s2 conv1 512->256", (8, 128, 128), 512, 256, 1, 1, False),
    ("s2 conv3 256->512 +res", (8, 128, 128), 256, 512, 1, 1, True),
    ("s3 conv1 1024->512", (8, 64, 64), 1024, 512, 1, 1, False),
    ("s3 conv3 512->1024 +res", (8, 64, 64), 512, 1024, 1, 1, True),
    ("s4 conv3 1024->2048 +res", (8, 32, 32), 1024, 2048, 1, 1, True),
    ("aspp 3x3 d6 2048->256", (8, 32, 32), 2048, 256, 3, 1, False),
    ("mask 3x3 256->256 14x14 x800", (800, 14, 14), 256, 256, 3, 1, False),
]
ACCURACY = [
    ("3x3 64->96 ragged", (2, 19, 23), 64, 96, 3, 1, False, 1.0),
    ("1x1 256->128 +res", (2, 24, 24), 256, 128, 1, 1, True, 1.0),
    ("3x3 256->256", (1, 32, 32), 256, 256, 3, 1, False, 1.0),
    ("3x3 s2 128->128", (2, 33, 33), 128, 128, 3, 2, False, 1.0),
    ("1x1 2048->256 (long K)", (1, 16, 16), 2048, 256, 1, 1, False, 1.0),
    ("3x3 256->256, inputs x 1e-3", (1, 32, 32), 256, 256, 3, 1, False, 1e-3),
    ("3x3 256->256, inputs x 1e3", (1, 32, 32), 256, 256, 3, 1, False, 1e3),
]


def run(mode, x, dc, k, stride, r, out=None):
    ops.set_conv_math(mode)
    return ops.conv2d(x, dc, stride=stride, padding="same", act=_lib.ACT_NONE, residual=r, out=out)


def main():
    rng = np.random.default_rng(0)
    if "--no-accuracy" not in sys.argv:
        print("max |err| / max |ref| and rms err / rms ref against fp64 (same fp32 operands)")
        for label, (B, H, W), cin, cout, k, stride, res, scale in ACCURACY:
            xn = (rng.normal(size=(B, H, W, cin)) * scale).astype(np.float32)
            w = (rng.normal(size=(k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
            b = rng.normal(size=(cout,)).astype(np.float32) * np.float32(scale)
            x = torch.from_numpy(xn).cuda()
            dc = ops.DeviceConv(packing.pack_dense(w, b), "cuda")
            xd = torch.from_numpy(xn).double().permute(0, 3, 1, 2)
            wd = torch.from_numpy(w).double().permute(3, 2, 0, 1)
            Ho, Wo = -(-H // stride), -(-W // stride)
            pt = max((Ho - 1) * stride + k - H, 0)
            pw = max((Wo - 1) * stride + k - W, 0)
            xp = F.pad(xd, (pw // 2, pw - pw // 2, pt // 2, pt - pt // 2))
            ref = F.conv2d(xp, wd, torch.from_numpy(b).double(), stride=stride).permute(0, 2, 3, 1).numpy()
            r = None
            if res:
                rn = (rng.normal(size=ref.shape) * scale).astype(np.float32)
                r = torch.from_numpy(rn).cuda()
                ref = ref + rn.astype(np.float64)
            line = f"{label:34s}"
            for mode in ("f32", "f32x3", "f16"):
                got = run(mode, x, dc, k, stride, r).cpu().numpy().astype(np.float64)
                e = got - ref
                line += f"  {mode}: {np.abs(e).max() / np.abs(ref).max():.2e} / {np.sqrt((e * e).mean() / (ref * ref).mean()):.2e}"
            print(line, flush=True)
    print("launch times (best of 3 x 10)")
    reps = 10
    for label, (B, H, W), cin, cout, k, stride, res in TIMED:
        x = torch.from_numpy(rng.normal(size=(B, H, W, cin)).astype(np.float32)).cuda()
        w = (rng.normal(size=(k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
        b = rng.normal(size=(cout,)).astype(np.float32)
        r = torch.from_numpy(rng.normal(size=(B, H, W, cout)).astype(np.float32)).cuda() if res else None
        dc = ops.DeviceConv(packing.pack_dense(w, b), "cuda")
        gf = 2.0 * B * H * W * cin * cout * k * k / 1e9
        line = f"{label:32s} {gf:7.1f} GF"
        outs = {}
        for mode in ("f32", "f32x3"):
            outs[mode] = run(mode, x, dc, k, stride, r)
            best = []
            for _ in range(3):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(reps):
                    run(mode, x, dc, k, stride, r, out=outs[mode])
                e.record()
                torch.cuda.synchronize()
                best.append(s.elapsed_time(e) / reps)
            t = min(best)
            line += f"  {mode}: {t * 1e3:8.1f} us {gf / t:7.1f} TF"
        d = (outs["f32"] - outs["f32x3"]).abs().max().item() / outs["f32"].abs().max().item()
        print(line + f"  max diff/max {d:.1e}", flush=True)
    ops.set_conv_math("f32")


if __name__ == "__main__":
    main()
