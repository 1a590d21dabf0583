#!/usr/bin/env python3
"""The fp32-tensor ResNeXt stem: fused conv 7x7 s2 + ReLU + max-pool (csrc/stem_f32.hip; --math f32x3: stem_x3.hip) against the
two launches it replaces.
Usage (GPU box): python scripts/stem_bench.py [--lib experiment.so] [--reps 20]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instance-segmentation-road-project_amd")]
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None, help="an experiment build of the library (the product path has no override)")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--math", default="f32", choices=("f32", "f32x3"))
    args = ap.parse_args()
    from masklab_hip import _lib, ops, packing
    if args.lib:
        _lib.LIB_PATH = os.path.abspath(args.lib)
    ops.set_conv_math(args.math)
    rng = np.random.default_rng(0)
    w = (rng.normal(size=(7, 7, 3, 64)) * 0.08).astype(np.float32)
    b = rng.normal(size=(64,)).astype(np.float32)
    dc = ops.DeviceConv(packing.pack_rowspan(w, b), "cuda")
    for B, H, W in [(8, 1024, 1024), (16, 1280, 1280), (1, 1024, 1024), (1, 512, 512)]:
        x4 = torch.zeros((B, H, W, 4), device="cuda")
        x4[..., :3] = torch.from_numpy(rng.normal(size=(B, H, W, 3)).astype(np.float32)).cuda()

        def two():
            return ops.maxpool3x3s2(ops.conv2d(x4, dc, stride=2, padding=((3, 3), (3, 3)), act=_lib.ACT_RELU), pad=1)

        def one():
            return ops.stem_pool(x4, dc)

        same = bool(torch.equal(one(), two()))
        t = {}
        for name, fn in (("two launches", two), ("fused", one)):
            for _ in range(3):
                fn()
            best = 1e9
            for _ in range(3):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(args.reps):
                    fn()
                e.record()
                torch.cuda.synchronize()
                best = min(best, s.elapsed_time(e) / args.reps)
            t[name] = best
        gf = 2.0 * B * (H // 2) * (W // 2) * 64 * 147 / 1e9
        print(f"{B:3d} x {H}x{W}: two launches {1e3 * t['two launches']:8.1f} us | fused {1e3 * t['fused']:8.1f} us "
              f"{gf / t['fused']:6.1f} TF (147 taps) | x{t['two launches'] / t['fused']:.2f} | bit-identical {same}", flush=True)


if __name__ == "__main__":
    main()
