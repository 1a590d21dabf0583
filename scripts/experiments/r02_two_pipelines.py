#!/usr/bin/env python3
"""Experiment (round 2): does running the batch as TWO independent half-batch pipelines on two HIP streams beat one
full-batch pipeline?  Kernel boundaries and partly filled last rounds of one pipeline would be filled by the other.
Usage (GPU box, repo root): python scripts/experiments/r02_two_pipelines.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "instance-segmentation-road-project_amd"))
import numpy as np
import torch

import bench


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda:0")
    cfg, m1, _, _ = bench.build_model("resnext50", dev)
    _, m2, _, _ = bench.build_model("resnext50", dev)
    images = torch.from_numpy(np.random.default_rng(1234).integers(0, 256, (8, 1024, 1024, 3), dtype=np.uint8)).to(dev)
    a, b = images[:4].contiguous(), images[4:].contiguous()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

    def one():
        return m1(images)

    def two():
        cur = torch.cuda.current_stream()
        sa.wait_stream(cur)
        sb.wait_stream(cur)
        with torch.cuda.stream(sa):
            st_a = m1._stage1(a)
        with torch.cuda.stream(sb):
            st_b = m2._stage1(b)
        with torch.cuda.stream(sa):
            oa = m1._stage2(st_a)
        with torch.cuda.stream(sb):
            ob = m2._stage2(st_b)
        cur.wait_stream(sa)
        cur.wait_stream(sb)
        return oa, ob

    for name, fn in (("one pipeline, batch 8", one), ("two pipelines, batch 4 + 4", two), ("one pipeline, batch 8", one),
                     ("two pipelines, batch 4 + 4", two)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print(f"{name}: {dt * 1e3:.3f} ms per 8 images, {8 / dt:.1f} img/s", flush=True)


if __name__ == "__main__":
    main()
