# Experiment helper: the 8 x 1024^2 ResNeXt-50 forward eager / stage-1 hipGraph + eager stage 2 / whole-forward hipGraph, under both
# fp32-tensor conv maths (round 3: 21.18 / 21.87 / 21.66 ms fp32, 12.77 / 13.08 / 13.04 ms f32x3 -- eager with the auxiliary streams wins
# at this batch; the graphs pay off at batch 1).
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "instance-segmentation-road-project_amd")]
import numpy as np, torch
import bench
from masklab_hip import ops


def main():
    dev = torch.device("cuda", 0)
    cfg, model, w, hot = bench.build_model("resnext50", dev)
    images = torch.from_numpy(np.random.default_rng(1234).integers(0, 256, (8, 1024, 1024, 3), dtype=np.uint8)).to(dev)
    def run(label, steps=20):
        for _ in range(4): model(images)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(steps): model(images)
        torch.cuda.synchronize()
        print(f"{label:40s} {1e3 * (time.perf_counter() - t) / steps:7.3f} ms", flush=True)
    for math in ("f32", "f32x3"):
        ops.set_conv_math(math)
        model.enable_graphs(False); model.device_counts = "auto"
        run(f"{math} eager")
        model.enable_graphs(True); model.device_counts = False
        run(f"{math} stage-1 graph + eager stage 2")
        model.enable_graphs(True); model.device_counts = "auto"
        run(f"{math} whole graph")
        model.enable_graphs(False)
    ops.set_conv_math("f32")


if __name__ == "__main__":
    main()
