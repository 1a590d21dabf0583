#!/usr/bin/env python3
"""Time the grouped 3x3 kernels on the headline shapes (8 x 1024^2 ResNeXt-50).  EXPERIMENT_LIB selects an ablation build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "instance-segmentation-road-project_amd"))
import numpy as np, torch
from masklab_hip import ops, packing, _lib
if os.environ.get("EXPERIMENT_LIB"):
    _lib.LIB_PATH = os.environ["EXPERIMENT_LIB"]      # (read by this script only; the product path has no override)


def main():
    shapes = [(8, 256, 256, 128, 4, 1), (8, 256, 256, 256, 8, 2), (8, 128, 128, 256, 8, 1), (8, 128, 128, 512, 16, 2), (8, 64, 64, 512, 16, 1)]
    for B, H, W, C, c, s in shapes:
        x = torch.randn((B, H, W, C), device="cuda")
        k = (np.random.default_rng(0).normal(size=(3, 3, C, c)) * 0.1).astype(np.float32)
        wg = torch.from_numpy(packing.pack_grouped_mfma4(k, C // c)).cuda()
        for _ in range(3):
            ops.gconv3x3(x, wg, None, c, stride=s, padding=((1, 1), (1, 1)), act=_lib.ACT_RELU)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            y = ops.gconv3x3(x, wg, None, c, stride=s, padding=((1, 1), (1, 1)), act=_lib.ACT_RELU)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        nbytes = (x.numel() + y.numel()) * 4
        print(f"c={c:2d} s{s} {H}x{W}x{C}: {us:7.1f} us  {nbytes / us / 1e6:6.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
