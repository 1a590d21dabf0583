#!/usr/bin/env python3
"""One launch of each listed conv shape, the caches flushed in between (a 1 GiB fill), for a rocprofv3 --pmc pass:
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format rocpd -d OUT -o x -- python3 scripts/conv_shape_pmc.py [f16]
    python3 scripts/rocpd_pmc_dispatches.py OUT/.../x_results.db conv
prints the counters per dispatch in launch order next to the algorithmic bytes this script prints."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instance-segmentation-road-project_amd")]
import numpy as np
import torch

from masklab_hip import _lib, ops, packing

# (label, B, H, W, cin, cout, k, residual)
SHAPES = [
    ("c2 1x1 64->256 sc", 8, 256, 256, 64, 256, 1, False),
    ("c2 1x1 64->128", 8, 256, 256, 64, 128, 1, False),
    ("c2 1x1 128->256 +res", 8, 256, 256, 128, 256, 1, True),
    ("c2 1x1 256->128", 8, 256, 256, 256, 128, 1, False),
    ("c3 1x1 256->512 +res", 8, 128, 128, 256, 512, 1, True),
    ("c3 1x1 512->256", 8, 128, 128, 512, 256, 1, False),
    ("c4 1x1 512->1024 +res", 8, 64, 64, 512, 1024, 1, True),
    ("c4 1x1 1024->512", 8, 64, 64, 1024, 512, 1, False),
    ("P3 tower 3x3 128->128", 8, 128, 128, 128, 128, 3, False),
    ("mask 3x3 128->128 (720 rois)", 720, 14, 14, 128, 128, 3, False),
    ("decoder 3x3 160->128", 8, 128, 128, 160, 128, 3, False),
]


def main():
    half = len(sys.argv) > 1 and sys.argv[1] == "f16"
    if half:
        ops.set_conv_math("f16s")
    dt = torch.float16 if half else torch.float32
    es = 2 if half else 4
    rng = np.random.default_rng(0)
    flush = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
    for label, B, H, W, cin, cout, k, res in SHAPES:
        x = torch.from_numpy(rng.normal(size=(B, H, W, cin)).astype(np.float32)).cuda().to(dt)
        w = rng.normal(size=(k, k, cin, cout)).astype(np.float32) * 0.05
        dc = ops.DeviceConv(packing.pack_dense(w, np.zeros(cout, np.float32)), "cuda")
        r = torch.from_numpy(rng.normal(size=(B, H, W, cout)).astype(np.float32)).cuda().to(dt) if res else None
        out = torch.empty((B, H, W, cout), dtype=dt, device="cuda")
        if half:
            dc.wgt_h
        flush.fill_(1.0)
        torch.cuda.synchronize()
        ops.conv2d(x, dc, padding="same", act=_lib.ACT_RELU, residual=r, out=out)
        torch.cuda.synchronize()
        rd = es * (x.numel() + (out.numel() if res else 0) + w.size)
        print(f"{label:32s} algorithmic read {rd / 1e6:8.1f} MB  write {es * out.numel() / 1e6:8.1f} MB", flush=True)


if __name__ == "__main__":
    main()
