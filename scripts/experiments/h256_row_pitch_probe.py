#!/usr/bin/env python3
"""EXPERIMENT: is the half 1x1 GEMM limited by L2 channel conflicts of its power-of-two row pitch?  The same conv on an
input whose rows are K elements apart (2 KB at K = 1024) and on a channel slice of a wider buffer (K + 64 elements apart).
GPU box: python scripts/experiments/h256_row_pitch_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instance-segmentation-road-project_amd")]
import numpy as np
import torch
from masklab_hip import _lib, ops, packing

def main():
    ops.set_conv_math("f16s")
    rng = np.random.default_rng(0)
    for label, (B, H, W), cin, cout in [("conv1 1024->512", (16, 80, 80), 1024, 512), ("conv3 512->1024", (16, 80, 80), 512, 1024),
                                         ("s2 512->256", (16, 160, 160), 512, 256)]:
        w = (rng.normal(size=(1, 1, cin, cout)) / np.sqrt(cin)).astype(np.float32)
        for tile in (4, 5):
            dc = ops.DeviceConv(packing.pack_dense(w, np.zeros(cout, np.float32), tile=tile), "cuda")
            line = f"{label:18s} tile {tile}:"
            for pad in (0, 64, 8):
                xw = torch.from_numpy(rng.normal(size=(B, H, W, cin + pad)).astype(np.float16)).cuda()
                out = torch.empty((B, H, W, cout + pad), dtype=torch.float16, device="cuda")
                best = 1e9
                for _ in range(3):
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    for _ in range(10):
                        ops.conv2d(xw, dc, padding="same", act=_lib.ACT_RELU, out=out, out_coff=0)
                    e.record()
                    torch.cuda.synchronize()
                    best = min(best, s.elapsed_time(e) / 10)
                line += f"  row pitch K+{pad}: {1e3 * best:7.1f} us"
            print(line, flush=True)


if __name__ == "__main__":
    main()
