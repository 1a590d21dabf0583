#!/usr/bin/env python3
"""Micro-benchmark of the MFMA conv kernel on the shapes that dominate the ResNeXt-50 workload.
Usage (GPU box): python scripts/conv_bench.py [--reps 20]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instance-segmentation-road-project_amd")]
import numpy as np
import torch

from masklab_hip import _lib, ops, packing

# (label, B, H, W, cin, cout, k, stride, residual)
SHAPES = [
    ("c2 1x1 64->128", 8, 256, 256, 64, 128, 1, 1, False),
    ("c2 1x1 64->256 sc", 8, 256, 256, 64, 256, 1, 1, False),
    ("c2 1x1 128->256 +res", 8, 256, 256, 128, 256, 1, 1, True),
    ("c2 1x1 256->128", 8, 256, 256, 256, 128, 1, 1, False),
    ("c2 1x1 256->256 (c3b1)", 8, 256, 256, 256, 256, 1, 1, False),
    ("c3 1x1 256->512 +res", 8, 128, 128, 256, 512, 1, 1, True),
    ("c3 1x1 512->256", 8, 128, 128, 512, 256, 1, 1, False),
    ("c4 1x1 512->1024 +res", 8, 64, 64, 512, 1024, 1, 1, True),
    ("c4 1x1 1024->512", 8, 64, 64, 1024, 512, 1, 1, False),
    ("c5 1x1 1024->2048 +res", 8, 32, 32, 1024, 2048, 1, 1, True),
    ("c5 1x1 2048->1024", 8, 32, 32, 2048, 1024, 1, 1, False),
    ("P3 tower 3x3 128->128", 8, 128, 128, 128, 128, 3, 1, False),
    ("mask 3x3 128->128 (720 rois)", 720, 14, 14, 128, 128, 3, 1, False),
    ("decoder 3x3 160->128", 8, 128, 128, 160, 128, 3, 1, False),
]


def ab_pipe(reps, math="f32", big=False):
    """generic kernel (tile 1) vs the pipelined 1x1 kernel (tile 4), interleaved rounds in one process.  --math f32x3: the
    split-operand form of both (round 4); --big: the 16 x 1280^2 ResNeXt-101 shapes instead of 8 x 1024^2 ResNeXt-50."""
    rng = np.random.default_rng(0)
    ops.set_conv_math(math)
    shapes = SHAPES
    if big:
        shapes = [("s1 64->128", 16, 320, 320, 64, 128, 1, 1, False), ("s1 64->256 sc", 16, 320, 320, 64, 256, 1, 1, False),
                  ("s1 128->256 +res", 16, 320, 320, 128, 256, 1, 1, True), ("s1 256->128", 16, 320, 320, 256, 128, 1, 1, False),
                  ("s2 256->512 +res", 16, 160, 160, 256, 512, 1, 1, True), ("s2 512->256", 16, 160, 160, 512, 256, 1, 1, False),
                  ("s3 512->1024 +res", 16, 80, 80, 512, 1024, 1, 1, True), ("s3 1024->512", 16, 80, 80, 1024, 512, 1, 1, False)]
    for label, B, H, W, cin, cout, k, stride, res in shapes:
        if k != 1 or stride != 1:
            continue
        x = torch.from_numpy(rng.normal(size=(B, H, W, cin)).astype(np.float32)).cuda()
        w = rng.normal(size=(1, 1, cin, cout)).astype(np.float32) * 0.05
        dcs = {t: ops.DeviceConv(packing.pack_dense(w, np.zeros(cout, np.float32), tile=t), "cuda") for t in (1, 4)}
        r = torch.from_numpy(rng.normal(size=(B, H, W, cout)).astype(np.float32)).cuda() if res else None
        out = torch.empty((B, H, W, cout), device="cuda")
        best = {1: [], 4: []}
        for rnd_ in range(5):
            for t in (1, 4):
                for _ in range(2):
                    ops.conv2d(x, dcs[t], padding="same", act=_lib.ACT_RELU, residual=r, out=out)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(reps):
                    ops.conv2d(x, dcs[t], padding="same", act=_lib.ACT_RELU, residual=r, out=out)
                e.record()
                torch.cuda.synchronize()
                best[t].append(s.elapsed_time(e) / reps)
        gf = 2.0 * out.numel() * cin / 1e9
        mb = 4 * (x.numel() + out.numel() * (2 if res else 1)) / 1e6
        m1, m4 = np.median(best[1]), np.median(best[4])
        outs = [ops.conv2d(x, dcs[t], padding="same", act=_lib.ACT_RELU, residual=r) for t in (1, 4)]
        diff = float((outs[0] - outs[1]).abs().max().item())
        print(f"{label:28s} generic {1e3 * m1:7.1f} us {gf / m1:6.1f} TF | pipe {1e3 * m4:7.1f} us {gf / m4:6.1f} TF "
              f"{mb / m4:6.0f} GB/s | x{m1 / m4:.2f} | max diff {diff:.3g}", flush=True)
    ops.set_conv_math("f32")


def half_pipe(reps):
    """the pipelined 1x1 kernel on fp16 tensors (BASELINE config-5 shapes: 16 x 1280^2 ResNeXt-101): HBM GB/s"""
    rng = np.random.default_rng(0)
    ops.set_conv_math("f16s")
    shapes = [("s1 64->128", 16, 320, 64, 128, False), ("s1 64->256 sc", 16, 320, 64, 256, False),
              ("s1 128->256 +res", 16, 320, 128, 256, True), ("s1 256->128", 16, 320, 256, 128, False),
              ("s2 256->512 +res", 16, 160, 256, 512, True), ("s2 512->256", 16, 160, 512, 256, False),
              ("s3 512->1024 +res", 16, 80, 512, 1024, True), ("s3 1024->512", 16, 80, 1024, 512, False),
              ("s4 1024->2048 +res", 16, 40, 1024, 2048, True), ("s4 2048->1024", 16, 40, 2048, 1024, False)]
    for label, B, S, cin, cout, res in shapes:
        x = torch.from_numpy(rng.normal(size=(B, S, S, cin)).astype(np.float16)).cuda()
        w = rng.normal(size=(1, 1, cin, cout)).astype(np.float32) * 0.05
        dc = ops.DeviceConv(packing.pack_dense(w, np.zeros(cout, np.float32)), "cuda")
        r = torch.from_numpy(rng.normal(size=(B, S, S, cout)).astype(np.float16)).cuda() if res else None
        out = torch.empty((B, S, S, cout), dtype=torch.float16, device="cuda")
        for _ in range(3):
            ops.conv2d(x, dc, padding="same", act=_lib.ACT_RELU, residual=r, out=out)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            ops.conv2d(x, dc, padding="same", act=_lib.ACT_RELU, residual=r, out=out)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / reps
        gf = 2.0 * out.numel() * cin / 1e9
        mb = 2 * (x.numel() + out.numel() * (2 if res else 1)) / 1e6
        print(f"{label:24s} {1e3 * ms:8.1f} us {gf / ms:7.1f} TF/s {mb / ms:7.0f} GB/s ({mb / ms / 80:.0f} % of 8 TB/s)", flush=True)
    ops.set_conv_math("f32")


def half_heads(reps):
    """the generic kernel on fp16 tensors: the head convs of the 16 x 1280^2 ResNeXt-101 workload (3x3, 128 / 256 channels).
    Prints a checksum of the output (sum of the half bit patterns) so that two library builds can be compared for equality."""
    rng = np.random.default_rng(0)
    ops.set_conv_math("f16s")
    shapes = [("P3 tower 3x3 128->128", 16, 160, 160, 128, 128, 3), ("P4 tower 3x3 128->128", 16, 80, 80, 128, 128, 3),
              ("FPN P3 3x3 128->128", 16, 160, 160, 128, 128, 3), ("decoder 3x3 160->128", 16, 160, 160, 160, 128, 3),
              ("mask 3x3 128->128 (1600 rois)", 1600, 14, 14, 128, 128, 3), ("lateral 1x1 2048->128", 16, 40, 40, 2048, 128, 1)]
    for label, B, H, W, cin, cout, k in shapes:
        x = torch.from_numpy(rng.normal(size=(B, H, W, cin)).astype(np.float16)).cuda()
        w = rng.normal(size=(k, k, cin, cout)).astype(np.float32) * 0.05
        dc = ops.DeviceConv(packing.pack_dense(w, np.zeros(cout, np.float32)), "cuda")
        out = torch.empty((B, H, W, cout), dtype=torch.float16, device="cuda")
        for _ in range(3):
            ops.conv2d(x, dc, padding="same", act=_lib.ACT_RELU, out=out)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(reps):
                ops.conv2d(x, dc, padding="same", act=_lib.ACT_RELU, out=out)
            e.record()
            torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e) / reps)
        gf = 2.0 * out.numel() * k * k * cin / 1e9
        chk = int(out.view(torch.int16).to(torch.int64).sum().item())
        print(f"{label:32s} {1e3 * best:8.1f} us {gf / best:7.1f} TF/s   checksum {chk}", flush=True)
    ops.set_conv_math("f32")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--half-heads", action="store_true")
    ap.add_argument("--lib", default=None, help="an experiment build of the library (the product path has no override)")
    ap.add_argument("--ab-pipe", action="store_true")
    ap.add_argument("--half-pipe", action="store_true")
    ap.add_argument("--math", default="f32", choices=("f32", "f32x3"), help="conv math of --ab-pipe")
    ap.add_argument("--big", action="store_true", help="--ab-pipe on the 16 x 1280^2 ResNeXt-101 shapes")
    args = ap.parse_args()
    if args.lib:
        _lib.LIB_PATH = os.path.abspath(args.lib)
    if args.half_heads:
        return half_heads(args.reps)
    if args.ab_pipe:
        return ab_pipe(args.reps, args.math, args.big)
    if args.half_pipe:
        return half_pipe(args.reps)
    rng = np.random.default_rng(0)
    tot_ms, tot_gf = 0.0, 0.0
    for label, B, H, W, cin, cout, k, stride, res in SHAPES:
        x = torch.from_numpy(rng.normal(size=(B, H, W, cin)).astype(np.float32)).cuda()
        w = rng.normal(size=(k, k, cin, cout)).astype(np.float32) * 0.05
        dc = ops.DeviceConv(packing.pack_dense(w, np.zeros(cout, np.float32)), "cuda")
        r = torch.from_numpy(rng.normal(size=(B, H // stride, W // stride, cout)).astype(np.float32)).cuda() if res else None
        out = torch.empty((B, H // stride, W // stride, cout), device="cuda")
        for _ in range(3):
            ops.conv2d(x, dc, stride=stride, padding="same", act=_lib.ACT_RELU, residual=r, out=out)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(args.reps):
            ops.conv2d(x, dc, stride=stride, padding="same", act=_lib.ACT_RELU, residual=r, out=out)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / args.reps
        gf = 2.0 * out.numel() * k * k * cin / 1e9
        mb = 4 * (x.numel() + out.numel() * (2 if res else 1)) / 1e6
        tot_ms += ms
        tot_gf += gf
        print(f"{label:32s} {1e3 * ms:8.1f} us {gf / ms:7.1f} TF/s {mb / ms:8.0f} GB/s")
    print(f"{'TOTAL':32s} {1e3 * tot_ms:8.1f} us {tot_gf / tot_ms:7.1f} TF/s")


if __name__ == "__main__":
    main()
