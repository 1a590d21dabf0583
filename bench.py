#!/usr/bin/env python3
"""bench.py -- images/sec of the MaskLab inference hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

A "step" = one complete hot-path forward (preprocess, backbone, P6/P7,
FPN, cls/box towers, box decode, DetectionProposal, RoI crop, mask head, ASPP + decoder) over one
batch of synthetic 1024x1024 RGB images already resident in HBM, plus -- for N>1 -- the RCCL
all-gather of the fixed-capacity per-GPU detections.  Default workload = BASELINE.json configs[2]
(ResNeXt-50 full MaskLab, 8 images per GPU; N GPUs process 8*N images = configs[3] at N=8,
weak scaling).  Weights are random-init of that architecture (no network for checkpoints), with the
class logits widened so that the NMS and mask head carry their full load (<=100 RoIs / image).

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects:
  roofline     -- the dominant kernel (MFMA implicit-GEMM conv, 128x128 tile): algorithmic FLOP of
                  all its launches in one step / their summed duration, timed with HIP events on
                  the launch stream in an instrumented step; peak = 157.3 TFLOP/s (fp32 MFMA).
  cpu_baseline -- the NumPy oracle forward ("port", not TF-Keras) timed on the host cores on a
                  bounded sample (1 image of the same workload).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "instance-segmentation-road-project_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np
import torch

PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
PEAK_HBM_GBS = 8000.0

WORKLOADS = {
    # name: (backbone, per-GPU batch, H, W, heads)
    "resnext50_full_b8_1024": ("resnext50", 8, 1024, 1024, "full"),      # BASELINE configs[2]/[3]
    "mobilenet_fpn_aspp_b1_1024": ("mobilenet", 1, 1024, 1024, "full"),  # BASELINE configs[1] (heads on device too)
    "mobilenet_full_b1_512": ("mobilenet", 1, 512, 512, "full"),         # BASELINE configs[0] shape
    "resnext50_full_b2_256": ("resnext50", 2, 256, 256, "full"),         # quick functional check
    "resnext101_full_b16_1280_f32": ("resnext101", 16, 1280, 1280, "full"),  # BASELINE configs[4] shape, fp32 path
    "resnext101_full_b16_1280_f16": ("resnext101", 16, 1280, 1280, "full"),  # BASELINE configs[4]: fp16 MFMA path
    "resnext50_full_b8_1024_f16": ("resnext50", 8, 1024, 1024, "full"),      # configs[2] shape on the fp16 MFMA path
}
PEAK_F16_MFMA_TFLOPS = 2500.0     # dense fp16/bf16 MFMA peak (MI355X_MICROARCH.md), never the 2:1-sparsity figure


def build_model(backbone, device, seed=0, hot_cls=True):
    from masklab_hip import ModelConfiguration, retinamasklab as R
    cfg = ModelConfiguration()
    cfg.backbone.backbone_type = backbone
    _, model = R.construct_masklab_networks(cfg)
    w = model.init_weights(seed)
    if hot_cls:
        for k in w:
            if k.startswith("classification_sub_net/") and k.endswith("/output/kernel"):
                w[k] = (w[k] * 8.0).astype(np.float32)
    model.load_weights(w, device)
    return cfg, model, w


def measured_traffic(kernel_label):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC pass (profiles/*_traffic.json,
    produced by scripts/pmc_traffic.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same
    command, FETCH x2 gfx950 correction).  None when no such file travels with the repo."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
    if not files:
        return None
    want = {"conv_mfma_128x128": "conv_mfma_kernel<2, 2, 2, 2", "conv_mfma_128x64": "conv_mfma_kernel<2, 2, 2, 1",
            "conv_mfma_128x32": "conv_mfma_kernel<4, 1, 1, 1"}.get(kernel_label)
    if want is None:
        return None
    try:
        data = json.load(open(files[-1]))["kernels"]
        for k, v in data.items():
            if want in k:
                return round(v["hbm_bytes_per_launch"])
    except Exception:
        return None
    return None


def cpu_baseline(cfg, weights, img, gpu_outs=None, gpu_kept=None, max_seconds=60.0):
    """Oracle forward on the host (1 image = rank 0's first bench image).  Returns (cpu_baseline, parity) for the
    JSON line: the timing, and -- the oracle being the checker -- how the GPU outputs for that image compare."""
    from oracle import masklab as O
    from oracle import metrics as OM
    H, W = img.shape[1:3]
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    try:                                  # the oracle's heavy lifting is BLAS matmul: report the threads BLAS really uses
        from threadpoolctl import threadpool_info
        blas = [int(i.get("num_threads", 0)) for i in threadpool_info() if i.get("user_api") == "blas"]
        if blas:
            cores = max(blas)
    except Exception:
        pass
    t0 = time.perf_counter()
    want = O.inference_forward(cfg, weights, img, literal_groups=False, return_internals=True)
    dt = time.perf_counter() - t0
    cpu = {"value": round(1.0 / dt, 4), "unit": "images/sec", "cores": cores, "kind": "port",
           "sample": f"1 image {H}x{W}, full hot-path forward, NumPy/BLAS oracle (not TF-Keras), {dt:.1f}s"}
    parity = None
    if gpu_outs is not None:
        want, internals = want
        names = ["cls_pred", "loc_pred", "roi_boxes", "roi_masks", "seg_pred"]
        got = dict(zip(names, gpu_outs))
        ref = dict(zip(names, want))
        diffs = {}
        for n in ("cls_pred", "loc_pred", "seg_pred"):
            diffs[n] = float(np.abs(got[n].astype(np.float64) - ref[n]).max()) if got[n].shape == ref[n].shape else None
        # Detections are compared by IDENTITY (anchor, class), not by row: this benchmark's synthetic class logits are
        # scaled x8 so that NMS and the mask head run at full load, which saturates many scores within 1e-6 of each
        # other -- the score ORDER of such near-ties legitimately depends on fp32 summation order.
        ref_kept = internals["kept"][:, 1:]                          # (anchor, class) of image 0, oracle order
        n_ref = len(ref_kept)
        g_kept = np.asarray(gpu_kept)[:n_ref] if gpu_kept is not None else np.zeros((0, 2), np.int64)
        ref_ids = {(int(a), int(c)): i for i, (a, c) in enumerate(ref_kept)}
        common = [(i, ref_ids[(int(a), int(c))]) for i, (a, c) in enumerate(g_kept) if (int(a), int(c)) in ref_ids]
        order_exact = bool(len(g_kept) == n_ref and np.array_equal(g_kept, ref_kept))
        mask_diff = 0.0
        box_diff = 0.0
        if common and got["roi_masks"].shape[2:] == ref["roi_masks"].shape[2:]:
            # roi_boxes / roi_masks rows are grouped by pyramid level; match rows through the box geometry + class
            def rows(t):
                bx = t["roi_boxes"][0]
                return {(round(float(r[0]), 2), round(float(r[1]), 2), round(float(r[2]), 2), round(float(r[3]), 2), int(r[4])): i
                        for i, r in enumerate(bx) if r[4] >= 0}
            rg, rr = rows(got), rows(ref)
            both = [k for k in rg if k in rr]
            if both:
                mask_diff = max(float(np.abs(got["roi_masks"][0, rg[k]].astype(np.float64) - ref["roi_masks"][0, rr[k]]).max())
                                for k in both)
                box_diff = max(float(np.abs(got["roi_boxes"][0, rg[k], 5] - ref["roi_boxes"][0, rr[k], 5])) for k in both)
            diffs["roi_masks(matched rows)"] = mask_diff
            diffs["roi_boxes.score(matched rows)"] = box_diff
        # precision / recall / F-measure at IoU 0.5 (reference engine/metrics.py:109-165) of the GPU
        # detections against the oracle's -- the stand-in for "box AP vs Keras ref", 1.0 = same detections
        pr, rc, fm = OM.detection_iou_metric(got["roi_boxes"], ref["roi_boxes"])
        frac_common = len(common) / max(n_ref, 1)
        parity = {"image": "rank 0, image 0 of the bench batch", "tolerance": 1e-3,
                  "max_abs_diff": {k: (None if v is None else float(f"{v:.3e}")) for k, v in diffs.items()},
                  "detections": n_ref, "same_detections": len(common), "order_exact": order_exact,
                  "detection_precision": round(float(pr[0]), 6), "detection_recall": round(float(rc[0]), 6),
                  "detection_fmeasure": round(float(fm[0]), 6),
                  "ok": bool(all(v is not None and v <= 1e-3 for v in diffs.values()) and frac_common >= 0.98
                             and float(fm[0]) >= 0.98)}
    return cpu, parity


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="resnext50_full_b8_1024", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--dump-launches", default=None, help="write one line per profiled launch to this file")
    ap.add_argument("--host-inputs", action="store_true",
                    help="every step starts from pinned HOST uint8 images (PCIe-inclusive rate; not the headline)")
    ap.add_argument("--graph", action="store_true",
                    help="replay stage 1 of the forward from a hipGraph (launch-bound small batches)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    # MASKLAB_BENCH_REHEARSAL=1: rehearse the N>1 code path on a ONE-GPU box -- every rank on cuda:0, gloo
    # instead of RCCL.  The numbers of such a run mean nothing; the driver's real runs never set it.
    rehearsal = os.environ.get("MASKLAB_BENCH_REHEARSAL") == "1"
    if rehearsal:
        import faulthandler
        faulthandler.enable()
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from masklab_hip import _lib, ops, parallel
    _lib.check(_lib.load().ml_device_check(), "ml_device_check")
    backbone, B, H, W, _heads = WORKLOADS[args.workload]
    f16 = args.workload.endswith("_f16")
    if f16:
        ops.set_conv_math("f16")       # dense convs: fp16 operands, fp32 accumulate; tensors stay fp32 in HBM
    cfg, model, weights = build_model(backbone, device)
    images = torch.from_numpy(np.random.default_rng(1234 + rank).integers(0, 256, (B, H, W, 3), dtype=np.uint8)).to(device)
    if args.graph:
        model.enable_graphs(True)
    if os.environ.get("MASKLAB_SIDE_STREAM") == "0":       # A/B knob; the model's default is on
        model.use_side_stream = False

    def mark(msg):
        if rehearsal:
            print(f"[rehearsal rank {rank}] {msg}", file=sys.stderr, flush=True)

    mark("model built")

    host_images = images.cpu().pin_memory() if args.host_inputs else None

    def step(collective=True):
        outs = model(host_images.to(device, non_blocking=True) if args.host_inputs else images)
        mark("forward enqueued")
        if world > 1 and collective:
            det = model.last_detections
            parallel.all_gather_detections(det["proposed"], det["counts"])
            mark("detections gathered")
        return outs

    step()                                 # untimed priming step, whatever --warmup is: module load, kernel attributes,
    torch.cuda.synchronize(device)         # scratch buffers, allocator pools and the RCCL communicator are one-time costs
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(device)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(device)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    roofline = None
    per_kernel = None
    if rank == 0 and not args.no_roofline:
        ops.PROFILE = []
        step(collective=False)          # rank 0 only: the other ranks are not in this step, so no collective
        torch.cuda.synchronize(device)
        recs, ops.PROFILE = ops.PROFILE, None
        if args.dump_launches:
            with open(args.dump_launches, "w") as f:
                for r in recs:
                    ms = r["start"].elapsed_time(r["end"])
                    f.write(f"{r['kernel']:28s} {r.get('shape', ''):60s} {1e3 * ms:9.1f} us "
                            f"{r['flops'] / 1e9 / max(ms, 1e-9):8.2f} TF/s {r['bytes'] / 1e6 / max(ms, 1e-9):9.1f} GB/s\n")
        agg = {}
        for r in recs:
            a = agg.setdefault(r["kernel"], {"launches": 0, "ms": 0.0, "gflop": 0.0, "mbytes": 0.0})
            a["launches"] += 1
            a["ms"] += r["start"].elapsed_time(r["end"])
            a["gflop"] += r["flops"] / 1e9
            a["mbytes"] += r["bytes"] / 1e6
        per_kernel = {k: {"launches": v["launches"], "ms": round(v["ms"], 3), "gflop": round(v["gflop"], 2),
                          "mbytes": round(v["mbytes"], 1),
                          "tflops": round(v["gflop"] / max(v["ms"], 1e-9), 2),
                          "gbs": round(v["mbytes"] / max(v["ms"], 1e-9), 1)} for k, v in agg.items()}
        dom = max(agg, key=lambda k: agg[k]["ms"])
        d = agg[dom]
        # the committed PMC pass was taken on the default workload only
        traffic = measured_traffic(dom) if args.workload == "resnext50_full_b8_1024" else None
        if dom.startswith("conv_mfma") and f16:
            # on the fp16 path the same kernel is bound by moving its fp32 operands, not by the matrix cores
            ach = d["mbytes"] / d["ms"]
            roofline = {"kernel": dom, "bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS,
                        "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": None,
                        "mfma_tflops": round(d["gflop"] / d["ms"], 2), "mfma_peak": PEAK_F16_MFMA_TFLOPS,
                        "algorithmic_bytes_per_launch": round(1e6 * d["mbytes"] / d["launches"]),
                        "launches_per_step": d["launches"], "avg_launch_us": round(1e3 * d["ms"] / d["launches"], 2),
                        "algorithmic_gflop_per_step": round(d["gflop"], 2)}
        elif dom.startswith("conv_mfma"):
            ach = d["gflop"] / d["ms"]          # GFLOP/ms = TFLOP/s
            roofline = {"kernel": dom, "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
                        "algorithmic_bytes_per_launch": round(1e6 * d["mbytes"] / d["launches"]),
                        "launches_per_step": d["launches"], "avg_launch_us": round(1e3 * d["ms"] / d["launches"], 2),
                        "algorithmic_gflop_per_step": round(d["gflop"], 2)}
        else:
            ach = d["mbytes"] / d["ms"]         # MB/ms = GB/s
            roofline = {"kernel": dom, "bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS,
                        "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": traffic,
                        "launches_per_step": d["launches"], "avg_launch_us": round(1e3 * d["ms"] / d["launches"], 2)}

    cpu = parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        first = images[:1].contiguous()
        gpu_first = None
        gpu_kept = None
        if not f16:                     # the fp16 MFMA mode has its own (looser) bar: tests/test_gpu_f16.py
            gpu_first = [o.cpu().numpy() for o in model(first, want_kept=True)]
            det0 = model.last_detections
            gpu_kept = det0["kept"][0, :int(det0["counts"][0])].cpu().numpy()
        cpu, parity = cpu_baseline(cfg, weights, first.cpu().numpy(), gpu_first, gpu_kept)

    if rank == 0:
        thr = cfg.detection.min_confidence
        outs = model(images)                    # the full bench batch again (the parity leg ran a single image)
        det = model.last_detections
        n_det = det["counts"].cpu().tolist() if det else []
        n_cand = (outs[0] >= thr).sum(dim=(1, 2)).cpu().tolist() if det else []
        total_images = B * world * args.steps
        line = {
            "metric": "images/sec at 1024x1024 (MaskLab inference hot path, full forward)",
            "value": round(total_images / dt, 3),
            "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16 MFMA operands, f32 accumulate, f32 tensors" if f16 else "f32", "data": "synthetic",
            "config": {"workload": args.workload, "per_gpu_batch": B, "global_batch": B * world,
                       "height": H, "width": W, "parallelism": f"dp{world}", "hipgraph": bool(args.graph),
                       "inputs": "pinned host memory, copied every step" if args.host_inputs else "resident in HBM",
                       "weights": "random init (cls logits x8 so NMS / mask head run at full load)",
                       "detections_per_image_rank0": n_det, "nms_candidates_per_image_rank0": n_cand},
            "roofline": roofline, "cpu_baseline": cpu, "parity": parity, "kernels": per_kernel,
        }
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
