#!/usr/bin/env python3
"""A handful of launches of ONE conv kernel class on its dominant shapes -- the program scripts/kernel_pmc.sh runs under
rocprofv3 --pmc.  GPU box: python3 scripts/conv_single.py f32|f32x3|f16s [launches per shape]
  f32   : conv_mfma_kernel<2,2,2,2,F32>  (128 x 128 tile)  -- the headline's dominant class
  f32x3 : conv_mfma_kernel<8,1,1,4,X3,3> (256 x 128 tile)  -- the split-operand mode's dominant class
  f16s  : conv_mfma_kernel<2,2,2,2,F16S> (128 x 128 tile, half tensors) -- the head convs of the fp16-storage mode (its #2 class;
          shapes then: the P3 tower conv at 16 x 160 x 160 and the decoder conv 160 -> 128)
Shapes (8 x 1024^2 ResNeXt-50): the P3 tower 3x3 conv 128 -> 128 at 128 x 128, the stage-4 1x1 conv 1024 -> 512 at 64 x 64."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instance-segmentation-road-project_amd")]
import numpy as np
import torch

from masklab_hip import _lib, ops, packing


def main():
    math = sys.argv[1] if len(sys.argv) > 1 else "f32"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    ops.set_conv_math(math)
    rng = np.random.default_rng(0)
    half = math == "f16s"
    shapes = (((16, 160, 160), 3, 128, 128), ((16, 160, 160), 3, 192, 128)) if half else \
             (((8, 128, 128), 3, 128, 128), ((8, 64, 64), 1, 1024, 512))
    for (B, H, W), k, cin, cout in shapes:
        x = torch.from_numpy(rng.normal(size=(B, H, W, cin)).astype(np.float16 if half else np.float32)).cuda()
        w = (rng.normal(size=(k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
        dc = ops.DeviceConv(packing.pack_dense(w, np.zeros(cout, np.float32), tile=1), "cuda")
        out = torch.empty((B, H, W, cout), device="cuda", dtype=torch.float16 if half else torch.float32)
        for _ in range(n):
            ops.conv2d(x, dc, padding="same", act=_lib.ACT_RELU, out=out)
        torch.cuda.synchronize()
    print("done")


if __name__ == "__main__":
    main()
