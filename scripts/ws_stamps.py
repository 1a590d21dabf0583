#!/usr/bin/env python3
"""Debug: per-chunk s_memtime stamps of the warp-specialised conv kernel (lib built with -DMASKLAB_STAMPS)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instance-segmentation-road-project_amd")]
import numpy as np, torch
from masklab_hip import _lib, ops, packing
lib = _lib.load(); raw = C.CDLL(_lib.LIB_PATH)
B, H, W, cin, cout, k = [int(v) for v in os.environ.get("STAMP_SHAPE", "8,32,32,2048,1024,1").split(",")]
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.normal(size=(B, H, W, cin)).astype(np.float32)).cuda()
dc = ops.DeviceConv(packing.pack_dense(rng.normal(size=(k, k, cin, cout)).astype(np.float32) * 0.05, np.zeros(cout, np.float32)), "cuda")
for _ in range(3):
    ops.conv2d(x, dc, padding="same", act=_lib.ACT_RELU)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (8 * 64))(); assert raw.ml_debug_read_stamps(buf) == 0
st = np.array(buf[:], dtype=np.uint64).reshape(8, 64).astype(np.int64)
n = 40
print("consumer: mfma block", np.median(st[1, 2:n] - st[0, 2:n]), " barrier wait", np.median(st[2, 2:n] - st[1, 2:n]), " chunk period", np.median(st[0, 3:n] - st[0, 2:n-1]))
print("producer: issue", np.median(st[4, 2:n] - st[3, 2:n]), " vmcnt wait", np.median(st[5, 2:n] - st[4, 2:n]), " period", np.median(st[3, 3:n] - st[3, 2:n-1]))
print("producer start relative to consumer chunk start:", (st[3, 2:12] - st[0, 2:12]).tolist())
print("producer ready relative to consumer mfma done:", (st[5, 2:12] - st[1, 2:12]).tolist())
