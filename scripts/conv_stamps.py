#!/usr/bin/env python3
"""Debug: in-kernel s_memtime stamps of the conv K loop (needs scripts/_ab/lib_stamps.so built with
-DMASKLAB_STAMPS and MASKLAB_HIP_LIB pointing at it).  Prints per-iteration cycle deltas of block 16 wave 0."""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instance-segmentation-road-project_amd")]
import numpy as np, torch
from masklab_hip import _lib, ops, packing

lib = _lib.load()
raw = C.CDLL(_lib.LIB_PATH)
rng = np.random.default_rng(0)
B, H, W, cin, cout, k = [int(v) for v in os.environ.get("STAMP_SHAPE", "8,128,128,128,128,3").split(",")]
x = torch.from_numpy(rng.normal(size=(B, H, W, cin)).astype(np.float32)).cuda()
dc = ops.DeviceConv(packing.pack_dense(rng.normal(size=(k, k, cin, cout)).astype(np.float32) * 0.05, np.zeros(cout, np.float32)), "cuda")
for _ in range(3):
    ops.conv2d(x, dc, padding="same", act=_lib.ACT_RELU)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (8 * 64))()
assert raw.ml_debug_read_stamps(buf) == 0
st = np.array(buf[:], dtype=np.uint64).reshape(8, 64).astype(np.int64)
names = ["loop top -> loads issued", "loads issued -> MFMA block done", "MFMA done -> (wait) LDS store issued",
         "store -> barrier passed", "barrier -> next loop top"]
n = max(3, min(36, k * k * ((cin + 31) // 32)))
print("iteration totals (s_memtime ticks, 100 MHz => x ~24 for shader cycles at 2.4 GHz):")
d = [st[1, :n] - st[0, :n], st[2, :n] - st[1, :n], st[3, :n] - st[2, :n], st[4, :n] - st[3, :n]]
tot = st[0, 1:n] - st[0, :n - 1]
for nm, v in zip(names, d):
    print(f"{nm:40s} median {np.median(v[1:n-1]):8.1f}  min {v[1:n-1].min():6d} max {v[1:n-1].max():6d}")
w = st[5, :n] - st[2, :n]
print(f"{'  of which: vmcnt(0) wait':40s} median {np.median(w[1:n-1]):8.1f}  min {w[1:n-1].min():6d} max {w[1:n-1].max():6d}")
print(f"{'whole iteration':40s} median {np.median(tot):8.1f}")
print("first 8 iterations:", [int(v) for v in tot[:8]])
e = st[7, :6]
print("kernel phases (cycles): entry->setup %d, first chunk load+store+barrier %d, K loop %d, epilogue issue %d, stores drained %d"
      % (e[1]-e[0], e[2]-e[1], e[3]-e[2], e[4]-e[3], e[5]-e[4]))
f = st[7]
print("setup detail: problem lookup %d, (to row setup) %d, row coords+offsets %d, rest of setup %d" % (f[8]-f[0], f[9]-f[8], f[10]-f[9], f[1]-f[10]))
print("epilogue detail: acc->LDS %d, barrier %d, read+bias+act+store %d" % (f[11]-f[3], f[12]-f[11], f[4]-f[12]))
