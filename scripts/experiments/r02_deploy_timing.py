#!/usr/bin/env python3
"""Time the deploy / serving wrappers around the inference model at 8 x 1024^2 ResNeXt-50."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "instance-segmentation-road-project_amd"))
import numpy as np, torch
import bench
from masklab_hip import retinamasklab as R, ops

def main():
    dev = torch.device("cuda:0")
    cfg, model, w, hot = bench.build_model("resnext50", dev)
    images = torch.from_numpy(np.random.default_rng(1234).integers(0, 256, (8, 1024, 1024, 3), dtype=np.uint8)).to(dev)
    deploy = R.construct_deploy_network(cfg, model)
    serving = R.construct_serving_network(cfg, deploy)

    def timeit(name, fn, n=10):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        print(f"{name}: {dt*1e3:.2f} ms per batch of 8 ({8/dt:.1f} img/s)", flush=True)

    timeit("inference model", lambda: model(images))
    timeit("deploy model   ", lambda: deploy(images))
    timeit("serving model  ", lambda: serving(images))
    ops.PROFILE = []
    deploy(images); torch.cuda.synchronize()
    recs, ops.PROFILE = ops.PROFILE, None
    agg = {}
    for r in recs:
        agg.setdefault(r["kernel"], [0, 0.0]); agg[r["kernel"]][0] += 1; agg[r["kernel"]][1] += r["start"].elapsed_time(r["end"])
    for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"  {k:28s} x{n:3d} {ms:.3f} ms")

    # where the serving wrapper's extra time goes: wall time of each stage with a sync in between
    import itertools
    def stage(name, fn):
        torch.cuda.synchronize(); t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize()
        print(f"  stage {name}: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True); return out
    for _ in range(2):
        det, ins, seg = stage("deploy", lambda: deploy(images))
        print("   shapes", tuple(det.shape), tuple(ins.shape), tuple(seg.shape))
        cp = stage("crop_and_pad", lambda: serving.crop_and_pad([images, det, ins, seg]))
        print("   crop_and_pad ->", tuple(cp.shape), cp.dtype)
        out = stage("summary", lambda: serving.summary([det, seg, cp]))
    ops.PROFILE = []
    serving.summary([det, seg, cp]); serving.crop_and_pad([images, det, ins, seg]); torch.cuda.synchronize()
    recs, ops.PROFILE = ops.PROFILE, None
    for r in recs:
        print(f"  {r['kernel']:28s} {r['start'].elapsed_time(r['end']):.3f} ms {r['shape']}")

    print("fused path:")
    for _ in range(3):
        stage("summary(from_rois)", lambda: serving.summary([det, seg, ins], from_rois=True))
    ops.PROFILE = []
    serving.summary([det, seg, ins], from_rois=True); torch.cuda.synchronize()
    recs, ops.PROFILE = ops.PROFILE, None
    for r in recs:
        print(f"  {r['kernel']:28s} {r['start'].elapsed_time(r['end']):.3f} ms {r['shape']}")
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        serving.summary([det, seg, ins], from_rois=True); torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=12, max_name_column_width=60))


if __name__ == "__main__":
    main()
