#!/usr/bin/env python3
"""A handful of launches of conv1x1_h256_kernel on the two ResNeXt-101 stage-3 shapes of the 16 x 1280^2 workload (the
dominant kernel of BASELINE configs[4]) -- the program scripts/h256_pmc.sh runs under rocprofv3 --pmc.
GPU box: python3 scripts/h256_single.py [launches per shape]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instance-segmentation-road-project_amd")]
import numpy as np
import torch

from masklab_hip import _lib, ops, packing


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    ops.set_conv_math("f16s")
    rng = np.random.default_rng(0)
    for (B, H, W), cin, cout, res in (((16, 80, 80), 1024, 512, False), ((16, 80, 80), 512, 1024, True)):
        x = torch.from_numpy(rng.normal(size=(B, H, W, cin)).astype(np.float16)).cuda()
        w = (rng.normal(size=(1, 1, cin, cout)) / np.sqrt(cin)).astype(np.float32)
        r = torch.from_numpy(rng.normal(size=(B, H, W, cout)).astype(np.float16)).cuda() if res else None
        dc = ops.DeviceConv(packing.pack_dense(w, np.zeros(cout, np.float32), tile=5), "cuda")
        out = torch.empty((B, H, W, cout), dtype=torch.float16, device="cuda")
        for _ in range(n):
            ops.conv2d(x, dc, padding="same", act=_lib.ACT_RELU, residual=r, out=out)
        torch.cuda.synchronize()
    print("done")


if __name__ == "__main__":
    main()
