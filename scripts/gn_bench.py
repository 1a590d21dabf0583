#!/usr/bin/env python3
"""Per-launch timing of the chunk-wise GroupNormalization kernels (csrc/groupnorm.hip) on the shapes of the ResNeXt workloads:
the five pyramid levels of a tower depth in one launch pair, the three RoI levels, one big map.
GPU box: python scripts/gn_bench.py [--lib experiment.so]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instance-segmentation-road-project_amd")]
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None, help="an experiment build of the library (the product path has no override)")
    ap.add_argument("--reps", type=int, default=30)
    args = ap.parse_args()
    from masklab_hip import _lib, ops
    if args.lib:
        _lib.LIB_PATH = os.path.abspath(args.lib)
    cases = [("towers 8 x 1024^2 fp32 (5 levels, C=128, G=16)", torch.float32, [(8, s, s, 128) for s in (128, 64, 32, 16, 8)], 16),
             ("mask head 800 RoIs fp32 (3 levels 14x14, C=128, G=16)", torch.float32, [(270, 14, 14, 128)] * 3, 16),
             ("decoder 8 x 128^2 fp32 (C=128, G=16)", torch.float32, [(8, 128, 128, 128)], 16),
             ("towers 16 x 1280^2 half (5 levels, C=128, G=16)", torch.float16, [(16, s, s, 128) for s in (160, 80, 40, 20, 10)], 16),
             ("decoder 16 x 160^2 half (C=128, G=16)", torch.float16, [(16, 160, 160, 128)], 16)]
    for label, dt, shapes, G in cases:
        probs = []
        for sh in shapes:
            x = torch.randn(sh, device="cuda", dtype=torch.float32).to(dt)
            probs.append(dict(x=x, gamma=torch.ones(sh[-1], device="cuda"), beta=torch.zeros(sh[-1], device="cuda"),
                              groups=G, eps=1e-5, relu=True, out=torch.empty_like(x)))
        for _ in range(3):
            ops.groupnorm_chunk_multi(probs)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                ops.groupnorm_chunk_multi(probs)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / args.reps)
        nbytes = sum(p["x"].numel() * p["x"].element_size() for p in probs)
        # statistics pass (1 read) + apply pass (1 read + 1 write)
        print(f"{label:58s} {1e3 * best:8.1f} us  {3 * nbytes / best / 1e9:6.2f} TB/s (stats + apply: 3 passes over {nbytes / 1e6:.0f} MB)", flush=True)


if __name__ == "__main__":
    main()
