#!/usr/bin/env python3
"""Per-launch timing of the ResNeXt grouped 3x3 kernels (csrc/gconv_mfma4.hip) on the shapes of the two ResNeXt workloads:
8 x 1024^2 ResNeXt-50 (fp32 tensors) and 16 x 1280^2 ResNeXt-101 (half tensors).  Prints us per launch and the
algorithmic HBM rate (input once + output once).  GPU box: python scripts/gconv_bench.py [--dtype f16|f32|both]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instance-segmentation-road-project_amd")]
import numpy as np
import torch

from masklab_hip import _lib, ops, packing

# (label, B, H, W, C, c, stride)
SHAPES = {
    "f32": [("r50 s1 256^2 C128 c4", 8, 256, 256, 128, 4, 1), ("r50 s2b1 256^2 C256 c8 s2", 8, 256, 256, 256, 8, 2),
            ("r50 s2 128^2 C256 c8", 8, 128, 128, 256, 8, 1), ("r50 s3b1 128^2 C512 c16 s2", 8, 128, 128, 512, 16, 2),
            ("r50 s3 64^2 C512 c16", 8, 64, 64, 512, 16, 1)],
    "f16": [("r101 s1 320^2 C128 c4", 16, 320, 320, 128, 4, 1), ("r101 s2b1 320^2 C256 c8 s2", 16, 320, 320, 256, 8, 2),
            ("r101 s2 160^2 C256 c8", 16, 160, 160, 256, 8, 1), ("r101 s3b1 160^2 C512 c16 s2", 16, 160, 160, 512, 16, 2),
            ("r101 s3 80^2 C512 c16", 16, 80, 80, 512, 16, 1), ("r101 s4b1 80^2 C1024 c32 s2", 16, 80, 80, 1024, 32, 2),
            ("r101 s4 40^2 C1024 c32", 16, 40, 40, 1024, 32, 1), ("r50 s4 32^2 C1024 c32 (8 img)", 8, 32, 32, 1024, 32, 1)],
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="both", choices=["f16", "f32", "both"])
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--lib", default=None, help="an experiment build of the library (the product path has no override)")
    args = ap.parse_args()
    if args.lib:
        _lib.LIB_PATH = os.path.abspath(args.lib)
    rng = np.random.default_rng(0)
    for dt in (("f32", "f16") if args.dtype == "both" else (args.dtype,)):
        tdt = torch.float16 if dt == "f16" else torch.float32
        for label, B, H, W, C, c, s in SHAPES[dt]:
            x = torch.randn((B, H, W, C), device="cuda", dtype=torch.float32).to(tdt)
            k = (rng.normal(size=(3, 3, C, c)) * 0.1).astype(np.float32)
            wg = torch.from_numpy(packing.pack_grouped_mfma4(k, C // c)).cuda()
            for _ in range(3):
                y = ops.gconv3x3(x, wg, None, c, stride=s, padding=((1, 1), (1, 1)), act=_lib.ACT_RELU)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    y = ops.gconv3x3(x, wg, None, c, stride=s, padding=((1, 1), (1, 1)), act=_lib.ACT_RELU)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / args.reps)
            nbytes = x.element_size() * (x.numel() + y.numel())
            chk = int(y.view(torch.int16 if dt == "f16" else torch.int32).to(torch.int64).sum().item())   # equal across builds
            print(f"{dt} {label:30s} {1e3 * best:8.1f} us  {nbytes / best / 1e9:7.2f} TB/s (algorithmic)  checksum {chk}", flush=True)


if __name__ == "__main__":
    main()
