#!/usr/bin/env python3
"""A/B on half tensors: conv1x1_pipe_kernel<_Float16> (tile 4, 128 x 128 tiles) vs conv1x1_h256_kernel (tile 5, 256 x 256
tiles), interleaved rounds in one process, results compared element by element.  GPU box: python scripts/h256_ab.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instance-segmentation-road-project_amd")]
import numpy as np
import torch

from masklab_hip import _lib, ops, packing

# (label, M rows as B x H x W, cin, cout, residual)
SHAPES = [
    ("r101 s3 conv1 1024->512", (16, 80, 80), 1024, 512, False),
    ("r101 s3 conv3 512->1024 +res", (16, 80, 80), 512, 1024, True),
    ("r101 s2 conv1 512->256", (16, 160, 160), 512, 256, False),
    ("r101 s2 conv3 256->512 +res", (16, 160, 160), 256, 512, True),
    ("r101 s1 conv3 128->256 +res", (16, 320, 320), 128, 256, True),
    ("r101 s1 conv1 256->128 (not eligible: N)", (16, 320, 320), 256, 128, False),
    ("r101 s4 conv1 2048->1024", (16, 40, 40), 2048, 1024, False),
    ("r101 s4 conv3 1024->2048 +res", (16, 40, 40), 1024, 2048, True),
    ("r101 s3 sc 512->1024 (ragged M)", (3, 77, 81), 512, 1024, False),
    ("r50 s3 conv3 256->512 +res", (8, 128, 128), 256, 512, True),
]


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--lib":        # an experiment build (the product path has no override)
        _lib.LIB_PATH = os.path.abspath(sys.argv[2])
    ops.set_conv_math("f16s")
    rng = np.random.default_rng(0)
    reps = 10
    for label, (B, H, W), cin, cout, res in SHAPES:
        x = torch.from_numpy(rng.normal(size=(B, H, W, cin)).astype(np.float16)).cuda()
        w = (rng.normal(size=(1, 1, cin, cout)) / np.sqrt(cin)).astype(np.float32)
        b = rng.normal(size=(cout,)).astype(np.float32)
        r = torch.from_numpy(rng.normal(size=(B, H, W, cout)).astype(np.float16)).cuda() if res else None
        outs, times = {}, {}
        for t in (4, 5):
            key = {4: "pipe 128x128", 5: "h256"}[t]
            dc = ops.DeviceConv(packing.pack_dense(w, b, tile=t), "cuda")
            try:
                outs[t] = ops.conv2d(x, dc, padding="same", act=_lib.ACT_RELU, residual=r)
            except RuntimeError as e:
                print(f"{label:44s} tile {t}: {str(e)[-90:]}")
                break
            best = []
            for _ in range(3):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(reps):
                    ops.conv2d(x, dc, padding="same", act=_lib.ACT_RELU, residual=r, out=outs[t])
                e.record()
                torch.cuda.synchronize()
                best.append(s.elapsed_time(e) / reps)
            times[key] = min(best)
        gf = 2.0 * B * H * W * cin * cout / 1e9
        mb = 2.0 * (x.numel() + B * H * W * cout * (2 if res else 1) + cin * cout) / 1e6
        line = f"{label:44s}"
        for key, tm in times.items():
            line += f" | {key}: {1e3 * tm:6.1f} us {gf / tm:6.1f} TF {mb / tm:6.0f} GB/s"
        if 4 in outs and 5 in outs:
            d = (outs[4].float() - outs[5].float()).abs().max().item()
            line += f" | max diff {d:.3g}  nan {bool(torch.isnan(outs[5]).any())}"
            # race screen: a staged buffer read before its request landed shows as RARE wrong tiles -- repeat the launch
            # into fresh buffers (cold / warm caches, other data in LDS from the previous launch) and compare every time
            dc5 = ops.DeviceConv(packing.pack_dense(w, b, tile=5), "cuda")
            bad = 0
            for i in range(30):
                o = ops.conv2d(x, dc5, padding="same", act=_lib.ACT_RELU, residual=r)
                bad += int(not torch.equal(o, outs[4]))
            line += f" | repeat mismatches {bad}/30"
        print(line, flush=True)


if __name__ == "__main__":
    main()
