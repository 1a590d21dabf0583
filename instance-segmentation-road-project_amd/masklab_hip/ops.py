"""Operator launchers: torch tensors (device memory + streams only) -> libmasklab_hip.so.

Every function enqueues HIP kernels on torch's CURRENT stream and returns the output tensor.
Tensors are NHWC float32, contiguous.  A `(buffer, channel_offset)` pair addresses a channel
slice of a wider buffer so concat / add are fused into the producing kernel.
There is no torch compute and no CPU fallback here.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .packing import PackedConv, resolve_padding

import weakref

_ws_cache = {}
_ws_retired = []     # superseded scratch buffers stay allocated WHILE some model holds a captured hipGraph (which may
                     # replay with their addresses); released once the last graph owner has dropped its graphs
_graph_owners = weakref.WeakSet()


def graph_owner_registered(model):
    _graph_owners.add(model)


def graph_owner_released(model):
    """A model dropped its captured graphs: with no owner left nothing can replay with a retired buffer's address."""
    _graph_owners.discard(model)
    if not len(_graph_owners):
        _ws_retired.clear()

# When set to a list, every launcher appends {"kernel", "flops", "bytes", "start", "end"} with
# torch.cuda.Event pairs recorded on the launch stream (bench.py's roofline leg).  `bytes` /
# `flops` are ALGORITHMIC: inputs read once + outputs written once (+ weights once), 2*MACs.
PROFILE = None
# When set to a list, every dense-conv launch appends (shape label, K slices per problem) as the library reports them
# (ml_conv2d_launch_splits): tests use it to name the launches whose K sum is cut differently when the batch changes.
LAUNCH_LOG = None


class _Prof:
    def __init__(self, kernel, flops, nbytes, shape=""):
        self.on = PROFILE is not None
        if self.on:
            self.rec = {"kernel": kernel, "flops": float(flops), "bytes": float(nbytes), "shape": shape,
                        "start": torch.cuda.Event(enable_timing=True), "end": torch.cuda.Event(enable_timing=True)}

    def __enter__(self):
        if self.on:
            self.rec["start"].record()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.rec["end"].record()
            PROFILE.append(self.rec)
        return False


# Arithmetic of the dense (MFMA) convolutions: "f32" = exact fp32 products (the default; configs 1-4),
# "f16" = fp16 MFMA operands (rounded to fp16 on their way into LDS), fp32 accumulation, tensors stay fp32,
# "f16s" = the fp16 MFMA path of BASELINE config 5 with fp16 STORAGE: the ResNeXt body AND the heads keep activations
# and weights in IEEE half in HBM (the stem writes half; every conv, GroupNorm, resize, RoI crop reads and writes half;
# accumulation, statistics, bias and activation are fp32 with one rounding at the store); only what detect.hip reads
# and what the model returns -- cls_pred, loc_pred, roi_boxes, roi_masks, seg_pred -- is fp32.  Kernels follow the
# dtype of the tensor they are given: an fp32 tensor in this mode (MobileNet's body) runs like "f16".
# "f32x3" = fp32 tensors and fp32-grade arithmetic on the f16 matrix pipe (ML_MATH_F32X3): every operand is split into
# two halves (22 bits), every product is three f16 MFMAs with fp32 accumulation; 1x1 convs run on the generic kernel.
# Process-wide switch, read when a conv is launched; set it through set_conv_math().
CONV_MATH = "f32"
_MATH_CODE = {"f32": 0, "f16": 1, "f16s": 1, "f32x3": 3}   # per-launch code of fp32-tensor convs; half tensors select ML_MATH_F16S


def set_conv_math(mode):
    global CONV_MATH
    if mode not in _MATH_CODE:
        raise ValueError(f"conv math must be one of {sorted(_MATH_CODE)}, got {mode!r}")
    CONV_MATH = mode


def dtype_label():
    """The arithmetic type the dense-conv path computes in (bench.py's `dtype`)."""
    return {"f32": "f32", "f32x3": "f32 tensors, products as 3 x f16 MFMA on split operands (22-bit), f32 accumulate",
            "f16": "f16 MFMA operands, f32 accumulate, f32 tensors",
            "f16s": "f16 MFMA, f32 accumulate, f16 tensors in backbone body and heads (f32 predictions)"}[CONV_MATH]


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def half_storage():
    """True in the "f16s" mode: backbone bodies keep their tensors in IEEE half."""
    return CONV_MATH == "f16s"


def _require_dev(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"masklab_hip: `{name}` must be a CUDA/HIP torch tensor -- the product path "
                           f"runs only on the MI355X kernels (no CPU fallback)")
    if not t.is_contiguous():
        raise RuntimeError(f"masklab_hip: `{name}` must be contiguous NHWC")


def workspace(nbytes, device, tag="ws"):
    """Grow-only scratch buffer per (device, tag, stream); 256-byte aligned by the torch allocator.
    Keyed by the launch stream too: work enqueued on two streams must not share scratch memory.  A buffer that is
    outgrown is retired, never freed (graphs captured earlier replay with its address)."""
    key = (str(device), tag, torch.cuda.current_stream().cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _ws_retired.append(buf)
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


class DeviceConv:
    """A PackedConv uploaded to the GPU."""

    def __init__(self, packed: PackedConv, device):
        self.p = packed
        self.wgt = torch.from_numpy(np.ascontiguousarray(packed.wgt)).to(device)
        self.bias = None if packed.bias is None else torch.from_numpy(packed.bias).to(device)
        self._wgt_h = None
        self._wgt_x3 = None

    @property
    def wgt_x3(self):
        """The packed weights split for ML_MATH_F32X3 (include/masklab_hip.h): every 32-float chunk of a row becomes 32
        halves hi(w) followed by 32 halves 2^11 (w - hi(w)) -- same bytes, same strides; made on first use."""
        if self._wgt_x3 is None:
            w = np.ascontiguousarray(self.p.wgt, dtype=np.float32)
            rows, ktot = w.shape
            if ktot % 32:
                raise RuntimeError("f32x3: the packed row length must be a multiple of 32 floats")
            c = w.reshape(rows, ktot // 32, 32)
            hi = c.astype(np.float16)
            lo = ((c - hi.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
            both = np.ascontiguousarray(np.concatenate([hi, lo], axis=2))            # [rows, chunks, 64] halves
            self._wgt_x3 = torch.from_numpy(both.view(np.float32).reshape(rows, ktot)).to(self.wgt.device)
        return self._wgt_x3

    @property
    def span_pad_h(self):
        """K floats per tap of the half packing: the span rounded up to a 64-deep chunk (128 bytes per row)."""
        return -(-self.p.span // 64) * 64

    @property
    def wgt_h(self):
        """The packed weights rounded to IEEE half (fp16-storage convs), every tap padded to span_pad_h; made on first
        use."""
        if self._wgt_h is None:
            p = self.p
            if p.cpp_shift != 30 or p.group_cin_step:
                raise RuntimeError("fp16 storage: image (row-span) and grouped-window convs have no half packing")
            taps = p.KH * p.KW
            w = p.wgt.reshape(p.n_pad, taps, p.span_pad)[:, :, :p.span]
            wh = np.zeros((p.n_pad, taps, self.span_pad_h), np.float16)
            wh[:, :, :p.span] = w
            self._wgt_h = torch.from_numpy(np.ascontiguousarray(wh.reshape(p.n_pad, taps * self.span_pad_h))).to(self.wgt.device)
        return self._wgt_h


def _conv_desc(x, dc, stride=1, padding="same", dilation=1, act=_lib.ACT_NONE, residual=None, out=None,
               out_coff=0, in_coff=0, out_view=None, out_dtype=None, live=None, gn_partials=None):
    """Build the ml_conv2d_desc for one problem.  -> (desc, result tensor, profile record args).
    A float16 `x` selects the fp16-storage kernels (ML_MATH_F16S: half weights, half residual); the output is half
    unless the destination says otherwise (`out` / `out_view` tensor of dtype float32, out_dtype=torch.float32, or a
    sigmoid activation: the predictions the model returns are fp32).  out_dtype=torch.float16 on an fp32 `x` makes the
    generic kernel store half (the stem of an fp16-storage body)."""
    p = dc.p
    _require_dev(x, "x")
    half_in = x.dtype == torch.float16
    if out is not None:
        odt = out.dtype
    elif out_view is not None:
        odt = out_view[0].dtype
    elif out_dtype is not None:
        odt = out_dtype
    else:
        odt = torch.float16 if (half_in and act != _lib.ACT_SIGMOID) else torch.float32
    if odt not in (torch.float16, torch.float32):
        raise ValueError(f"conv2d: output dtype {odt} not supported")
    if half_in and residual is not None and residual.dtype != torch.float16:
        raise ValueError("conv2d: an fp16-storage conv takes a float16 residual")
    es_o = 2 if odt == torch.float16 else 4
    B, H, W, Cbuf = x.shape
    if p.cpp_shift != 30:
        if Cbuf != p.cin_buffer:
            raise ValueError(f"row-span conv expects a {p.cin_buffer}-channel padded image, got {Cbuf}")
    elif not p.group_cin_step and in_coff + p.span > Cbuf:
        raise ValueError(f"conv2d: input has {Cbuf} channels, kernel needs [{in_coff},{in_coff + p.span})")
    Ho, Wo, pt, pl = resolve_padding(H, W, p.kh_real, p.kw_real, stride, dilation, padding)
    co = p.cout // 4 if p.shuffle2x2 else p.cout
    oh, ow = (2 * Ho, 2 * Wo) if p.shuffle2x2 else (Ho, Wo)
    d = _lib.ConvDesc()
    if out_view is not None:
        vt, elem_off, vcs, vbs = out_view
        _require_dev(vt, "out_view")
        if p.shuffle2x2 or vcs < co or vbs < oh * ow * vcs:
            raise ValueError("conv2d: bad out_view")
        if elem_off + (B - 1) * vbs + oh * ow * vcs > vt.numel():
            raise ValueError("conv2d: out_view exceeds the destination tensor")
        ret = vt
        d.out = vt.data_ptr() + es_o * elem_off
        d.out_cstride, d.out_coff, d.out_bstride = vcs, 0, vbs
    else:
        if out is None:
            out = torch.empty((B, oh, ow, co), dtype=odt, device=x.device)
            out_coff = 0
        else:
            _require_dev(out, "out")
            if tuple(out.shape[:3]) != (B, oh, ow) or out.dtype != odt:
                raise ValueError(f"conv2d: out buffer {tuple(out.shape)} / {out.dtype} does not match {(B, oh, ow)} / {odt}")
        ret = out
        d.out = out.data_ptr()
        d.out_cstride, d.out_coff, d.out_bstride = out.shape[3], out_coff, 0
    x3 = CONV_MATH == "f32x3" and not half_in
    d.in_, d.wgt, d.bias = x.data_ptr(), (dc.wgt_h if half_in else (dc.wgt_x3 if x3 else dc.wgt)).data_ptr(), \
        (dc.bias.data_ptr() if dc.bias is not None else None)
    if residual is not None:
        _require_dev(residual, "residual")
        if tuple(residual.shape[:3]) != (B, Ho, Wo):
            raise ValueError("conv2d: residual spatial shape mismatch")
        d.residual, d.res_cstride, d.res_coff = residual.data_ptr(), residual.shape[3], 0
    d.B, d.H, d.W = B, H, W
    d.in_cstride, d.in_coff = Cbuf, in_coff
    d.span, d.span_pad, d.cpp_shift = p.span, (dc.span_pad_h if half_in else p.span_pad), p.cpp_shift
    d.Ho, d.Wo = Ho, Wo
    d.KH, d.KW, d.stride, d.dil, d.pad_t, d.pad_l = p.KH, p.KW, stride, dilation, pt, pl
    d.cout, d.n_pad = p.cout, p.n_pad
    d.act, d.group_cin_step, d.shuffle2x2, d.tile = act, p.group_cin_step, p.shuffle2x2, p.tile
    if gn_partials is not None:        # float64 [tiles, 4, 2]: the epilogue also sums what each 128-row tile stores (per wave)
        _require_dev(gn_partials, "gn_partials")
        if gn_partials.dtype != torch.float64 or gn_partials.numel() < 8 * ((B * Ho * Wo + 127) // 128):
            raise ValueError("conv2d: gn_partials must be a float64 tensor of 4 x 2 values per 128-row tile")
        if d.out_coff % 4 or d.out_cstride % 4:        # only the vector epilogue writes the partial sums
            raise ValueError("conv2d: gn_partials needs a destination slice aligned to 4 channels")
        d.gn_partials = gn_partials.data_ptr()
    if live is not None:               # (device int32 tensor [1], slots per image): a fixed-capacity RoI batch
        lv, period = live
        _require_dev(lv, "live")
        d.live, d.live_period = lv.data_ptr(), int(period)
    if half_in:
        d.math, d.out_f16 = 2, int(odt == torch.float16)          # ML_MATH_F16S
    else:
        d.math, d.out_f16 = _MATH_CODE[CONV_MATH], int(odt == torch.float16)
        if d.out_f16 and d.math != 1:
            raise ValueError("conv2d: a half output from fp32 input needs the fp16 MFMA mode (set_conv_math('f16s'))")
    M = B * Ho * Wo
    real_cin = p.span if p.cpp_shift == 30 else 3
    es_in, es_out = x.element_size(), (2 if odt == torch.float16 else 4)
    nbytes = (es_in * B * H * W * real_cin * (1 if not p.group_cin_step else p.n_pad // 32) + es_out * M * p.cout +
              es_in * p.cout * p.k_real + (es_out * M * p.cout if residual is not None else 0))
    shape = f"M={M} N={p.cout} K={p.KH * p.KW * d.span_pad} k{p.kh_real}x{p.kw_real} s{stride} d{dilation} HxW={H}x{W}"
    return d, ret, (2.0 * M * p.cout * p.k_real, nbytes, shape)


def _conv_kernel_name(p, descs=None, n=1, half=False):
    """Name of the kernel instantiation a launch runs on (profiling hook only).  With the descriptors the library is
    asked which N tile it will really use: small launches run on narrower tiles than the weights were packed for."""
    lib = _lib.load()
    bn = lib.ml_conv2d_launch_ntile(descs, n, 1) if descs is not None and PROFILE is not None else 0
    bm = (lib.ml_conv2d_launch_mtile(descs, n, 1) if bn else 0) or 128
    if not bn:
        bn = lib.ml_conv2d_ntile(p.cout, p.tile)
    return "conv_mfma_%dx%d%s%s" % (bm, bn, "_grouped" if p.group_cin_step else "",
                                     "_h" if half else {"f32": "", "f32x3": "_x3"}.get(CONV_MATH, "_f16"))


def gn_fusable(out_shape, C_out, groups, dc, launch_tiles, dtype):
    """Can the GroupNormalization behind this conv take its statistics from the conv's epilogue (ml_conv2d_desc.gn_partials)?
    out_shape = (B, Ho, Wo): whole 128-row tiles per image and per chunk, one 128-wide N tile, fp32, and a launch big
    enough that the library neither narrows its tiles nor cuts K (ml_conv2d_gn_min_launch_tiles(): 257 tiles of 128 x 128 in
    all on a 256-CU part).
    -> (sum, sum of squares) pairs per chunk (4 per tile: one per wave), or 0."""
    B, Ho, Wo = out_shape
    hw = Ho * Wo
    p = dc.p
    # fp32 tensors on the exact / split-operand products, or half tensors (the heads of the fp16-storage mode: the sums are
    # then of the ROUNDED values the conv stores, which is what that mode's GroupNorm statistics pass reads)
    math_ok = (CONV_MATH in ("f32", "f32x3") and dtype == torch.float32) or (CONV_MATH == "f16s" and dtype == torch.float16)
    ok = (math_ok and C_out == 128 and p.cout == 128 and p.n_pad == 128 and
          not p.shuffle2x2 and not p.group_cin_step and hw % 128 == 0 and hw % groups == 0 and (hw // groups) % 128 == 0 and
          launch_tiles >= _gn_min_launch_tiles())
    return 4 * ((hw // groups) // 128) if ok else 0


_GN_MIN_TILES = None


def _gn_min_launch_tiles():
    """The launch size from which the LIBRARY neither narrows the tiles nor cuts K on this device (asked once; not a
    constant of this file: it follows the device's compute-unit count)."""
    global _GN_MIN_TILES
    if _GN_MIN_TILES is None:
        _GN_MIN_TILES = int(_lib.load().ml_conv2d_gn_min_launch_tiles())
    return _GN_MIN_TILES


def _log_launch(arr, n, ws, label):
    if LAUNCH_LOG is not None:
        sp = (C.c_int32 * n)()
        _lib.check(_lib.load().ml_conv2d_launch_splits(arr, n, ws.numel(), sp), "ml_conv2d_launch_splits")
        LAUNCH_LOG.append((label, tuple(int(v) for v in sp)))


def conv2d(x, dc: DeviceConv, stride=1, padding="same", dilation=1, act=_lib.ACT_NONE,
           residual=None, out=None, out_coff=0, in_coff=0, out_view=None, out_dtype=None, gn_partials=None):
    """ml_conv2d_multi_f32 with one problem (split-K enabled through the shared workspace).
    `x` [B,H,W,Cbuf]; reads channels [in_coff, in_coff+cin).  Writes into
    `out[..., out_coff:out_coff+cout]` when given, else allocates.  `out_view=(tensor, elem_off,
    cstride, bstride)` writes image b's pixels at tensor.data + elem_off + b*bstride with row
    pitch cstride (used to land a level's head directly in the concatenated prediction)."""
    lib = _lib.load()
    if x.dtype == torch.float16 and stride == 2 and dc.p.kh_real == 1 and dc.p.kw_real == 1:
        # a strided 1x1 conv on half tensors = the stride-1 kernel on the sampled pixels (ResNext.py:199-203 shortcuts)
        x, stride = subsample2_h(x), 1
    d, ret, (flops, nbytes, shape) = _conv_desc(x, dc, stride, padding, dilation, act, residual, out, out_coff,
                                                in_coff, out_view, out_dtype, gn_partials=gn_partials)
    ws = workspace(lib.ml_conv2d_workspace_bytes(), x.device, "conv")
    name = _conv_kernel_name(dc.p, C.byref(d), 1, half=x.dtype == torch.float16)
    if PROFILE is not None:
        which = lib.ml_conv2d_uses_pipe(C.byref(d))          # 1: the 128 x 128 pipelined kernel, 2: the half 256 x 256 one
        if which:
            name = "conv1x1_h256_h" if which == 2 else (
                "conv1x1_pipe_h" if x.dtype == torch.float16 else ("conv1x1_pipe_x3" if CONV_MATH == "f32x3" else "conv1x1_pipe"))
    _log_launch(C.byref(d), 1, ws, shape)
    with _Prof(name, flops, nbytes, shape) as prof:
        if prof.on and name == "conv1x1_h256_h":
            # bytes this launch stages through the CUs' L1 -> LDS path: every 256-row tile takes its 256 x K activation rows
            # and the 256 x K weight rows of its N tile, 2 bytes each (what bounds that kernel: profiles/r04_h256_pmc.md)
            M, N, K = d.B * d.Ho * d.Wo, d.cout, d.span
            prof.rec["staged_bytes"] = float(-(-M // 256) * (N // 256) * 2 * 256 * K * 2)
        _lib.check(lib.ml_conv2d_multi_f32(C.byref(d), 1, _ptr(ws), ws.numel(), _stream()), "ml_conv2d_multi_f32")
    return ret


def conv2d_multi(problems):
    """One launch for several independent convs of the same tile shape.
    problems: list of dicts with keys x, dc and the keyword arguments of conv2d().  -> list of results."""
    lib = _lib.load()
    n = len(problems)
    if n == 0:
        return []
    if n > 12:
        return conv2d_multi(problems[:12]) + conv2d_multi(problems[12:])
    arr = (_lib.ConvDesc * n)()
    rets, flops, nbytes, shapes = [], 0.0, 0.0, []
    for i, pr in enumerate(problems):
        pr = dict(pr)
        d, ret, (f, nb, shape) = _conv_desc(pr.pop("x"), pr.pop("dc"), **pr)
        arr[i] = d
        rets.append(ret)
        flops += f
        nbytes += nb
        shapes.append(shape)
    name = _conv_kernel_name(problems[0]["dc"].p, arr, n, half=problems[0]["x"].dtype == torch.float16)
    ws = workspace(int(lib.ml_conv2d_workspace_bytes()), problems[0]["x"].device, "conv")
    _log_launch(arr, n, ws, " | ".join(shapes))
    with _Prof(name, flops, nbytes, f"multi x{n}"):
        _lib.check(lib.ml_conv2d_multi_f32(arr, n, _ptr(ws), ws.numel(), _stream()), "ml_conv2d_multi_f32")
    return rets


def deconv2x2_out1x1_multi(problems, ncls, act_mid, act_out):
    """ml_deconv2x2_out1x1_f32: Conv2DTranspose(2x2, s2) + act_mid -> Conv2D 1x1 + act_out, up to 4 RoI levels per launch.
    problems: dicts x [R,h,w,K] fp32 (R = images * rois_per_image), dc (DeviceConv of the transposed conv), wo_table,
    bo (device tensors from packing.pack_out1x1_table), out (the [B,total,2h,2w,ncls] tensor), out_base (elements),
    rois_per_image."""
    lib = _lib.load()
    n = len(problems)
    if n == 0:
        return
    if n > _lib.DECONV_OUT_MAX_PROBLEMS:
        deconv2x2_out1x1_multi(problems[:_lib.DECONV_OUT_MAX_PROBLEMS], ncls, act_mid, act_out)
        deconv2x2_out1x1_multi(problems[_lib.DECONV_OUT_MAX_PROBLEMS:], ncls, act_mid, act_out)
        return
    arr = (_lib.DeconvOutProblem * n)()
    flops = nbytes = 0.0
    K = cmid = cp = None
    for i, pr in enumerate(problems):
        x, dc, out = pr["x"], pr["dc"], pr["out"]
        _require_dev(x, "x")
        _require_dev(out, "out")
        if x.dtype not in (torch.float32, torch.float16) or not x.is_contiguous() or out.dtype != torch.float32:
            raise ValueError("deconv2x2_out1x1: x must be a contiguous fp32 / fp16 [R,h,w,K] tensor, out fp32")
        if x.dtype != problems[0]["x"].dtype:
            raise ValueError("deconv2x2_out1x1: the problems of one launch must share the storage type")
        R, h, w, Kx = x.shape
        p = dc.p
        half = x.dtype == torch.float16
        if not p.shuffle2x2 or p.span != Kx or p.span_pad != Kx or p.cout % 4 or p.n_pad != p.cout or (half and Kx % 64):
            raise ValueError("deconv2x2_out1x1: `dc` must be a packed Conv2DTranspose whose input width is a multiple of "
                             "32 (64 for fp16 tensors)")
        if K is None:
            K, cmid, cp = Kx, p.cout // 4, int(pr["wo_table"].shape[-1])
        elif (K, cmid, cp) != (Kx, p.cout // 4, int(pr["wo_table"].shape[-1])):
            raise ValueError("deconv2x2_out1x1: the problems of one launch must share K, C_mid and the table width")
        if tuple(pr["wo_table"].shape) != (cmid // 32, 16, 2, cp):
            raise ValueError("deconv2x2_out1x1: wo_table has the wrong shape")
        n_l = int(pr["rois_per_image"])
        d = arr[i]
        d.x, d.wd, d.bd = x.data_ptr(), (dc.wgt_h if half else dc.wgt).data_ptr(), (dc.bias.data_ptr() if dc.bias is not None else None)
        d.wo_table, d.bo, d.out = pr["wo_table"].data_ptr(), pr["bo"].data_ptr(), out.data_ptr()
        d.M, d.hw, d.w, d.rois_per_image, d.reserved0 = R * h * w, h * w, w, n_l, 0
        d.out_image_stride, d.out_base = out.stride(0), int(pr["out_base"])
        if pr.get("live") is not None:
            d.live = pr["live"].data_ptr()
        last = (R // n_l - 1) * out.stride(0) + int(pr["out_base"]) + n_l * 4 * h * w * ncls
        if R % n_l or last > out.numel():
            raise ValueError("deconv2x2_out1x1: the RoI block does not fit the output tensor")
        flops += 2.0 * R * h * w * (4 * cmid * Kx + 4 * cmid * ncls)
        nbytes += x.element_size() * (x.numel() + 4 * cmid * Kx) + 4.0 * 4 * R * h * w * ncls
    half = problems[0]["x"].dtype == torch.float16
    fn = lib.ml_deconv2x2_out1x1_f16 if half else lib.ml_deconv2x2_out1x1_f32
    with _Prof("deconv2x2_out1x1_h" if half else "deconv2x2_out1x1", flops, nbytes, f"multi x{n}"):
        _lib.check(fn(arr, n, K, cmid, ncls, cp, act_mid, act_out, _stream()), "ml_deconv2x2_out1x1")


def gconv3x3(x, wgt, bias, c, stride=1, padding=((1, 1), (1, 1)), act=_lib.ACT_NONE):
    """ml_gconv3x3_f32: ResNeXt grouped 3x3, wgt [C,9,c] (packing.pack_grouped_mfma4)."""
    lib = _lib.load()
    _require_dev(x, "x")
    B, H, W, Cc = x.shape
    Ho, Wo, pt, pl = resolve_padding(H, W, 3, 3, stride, 1, padding)
    half = x.dtype == torch.float16
    out = torch.empty((B, Ho, Wo, Cc), dtype=x.dtype, device=x.device)
    fn = lib.ml_gconv3x3_f16 if half else lib.ml_gconv3x3_f32
    with _Prof("gconv3x3_mfma4_h" if half else "gconv3x3_mfma4", 2.0 * B * Ho * Wo * Cc * 9 * c,
               x.element_size() * (x.numel() + out.numel()) + 4 * wgt.numel(),
               f"M={B * Ho * Wo} C={Cc} c={c} s{stride} HxW={H}x{W}"):
        _lib.check(fn(_ptr(x), _ptr(wgt), _ptr(bias), _ptr(out), B, H, W, Cc, c, Ho, Wo, stride, pt, pl, act, _stream()),
                   "ml_gconv3x3")
    return out


def dwconv3x3(x, wgt, bias, stride=1, padding="same", dilation=1, act=_lib.ACT_NONE, out=None, out_coff=0):
    lib = _lib.load()
    _require_dev(x, "x")
    B, H, W, Cc = x.shape
    Ho, Wo, pt, pl = resolve_padding(H, W, 3, 3, stride, dilation, padding)
    half = x.dtype == torch.float16
    if out is None:
        out = torch.empty((B, Ho, Wo, Cc), dtype=x.dtype, device=x.device)
        out_coff = 0
    elif out.dtype != x.dtype:
        raise ValueError("dwconv3x3: out must have the input's dtype")
    fn = lib.ml_dwconv3x3_f16 if half else lib.ml_dwconv3x3_f32
    with _Prof("dwconv3x3_h" if half else "dwconv3x3", 18.0 * B * Ho * Wo * Cc,
               x.element_size() * (x.numel() + B * Ho * Wo * Cc) + 4 * 9 * Cc):
        _lib.check(fn(_ptr(x), _ptr(wgt), _ptr(bias), _ptr(out), B, H, W, Cc, Cc, 0,
                      out.shape[3], out_coff, Ho, Wo, stride, dilation, pt, pl, act, _stream()), "ml_dwconv3x3")
    return out


def stem_pool_h(x4, dc: "DeviceConv"):
    """ml_stem7x7s2_pool_f16: the ResNeXt stem (7x7 stride-2 conv + folded BN + ReLU) and the 3x3 stride-2 max-pool behind
    it in ONE kernel, fp32 NHWC4 image in, half pooled map out -- the fp16-storage mode only (the un-pooled stem output is
    never written).  `dc` = the stem's row-span DeviceConv; its weights rounded to half are made on first use."""
    lib = _lib.load()
    _require_dev(x4, "x4")
    p = dc.p
    B, H, W, c4 = x4.shape
    if x4.dtype != torch.float32 or c4 != 4 or p.cpp_shift == 30 or p.KH != 7 or p.span_pad != 32 or p.cout != 64 or p.n_pad != 64:
        raise ValueError("stem_pool_h: needs the fp32 NHWC4 image and the 7x7 / 64-filter row-span stem packing")
    wh = getattr(dc, "_stem_wgt_h", None)
    if wh is None:
        wh = dc._stem_wgt_h = torch.from_numpy(np.ascontiguousarray(p.wgt, np.float32).astype(np.float16)).to(x4.device)
    Hc, Wc = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    Hp, Wp = (Hc + 2 - 3) // 2 + 1, (Wc + 2 - 3) // 2 + 1
    out = torch.empty((B, Hp, Wp, 64), dtype=torch.float16, device=x4.device)
    with _Prof("stem7x7s2_pool_h", 2.0 * B * Hc * Wc * 64 * 147, 16 * B * H * W + 2 * out.numel() + 2 * 64 * 224,
               f"B={B} HxW={H}x{W} -> {Hp}x{Wp}x64"):
        _lib.check(lib.ml_stem7x7s2_pool_f16(_ptr(x4), _ptr(wh), _ptr(dc.bias), _ptr(out), B, H, W, Hp, Wp, _stream()),
                   "ml_stem7x7s2_pool_f16")
    return out


def stem_pool(x4, dc: "DeviceConv"):
    """ml_stem7x7s2_pool_f32 / _x3: the fp32-tensor twins of stem_pool_h, in the current conv math ("f32": exact fp32
    products, only those with a non-zero weight; "f32x3": the split-operand products) -- the same bits as
    conv2d(stem, relu) + maxpool3x3s2 in that math, without the un-pooled map in memory."""
    lib = _lib.load()
    _require_dev(x4, "x4")
    p = dc.p
    B, H, W, c4 = x4.shape
    if x4.dtype != torch.float32 or c4 != 4 or p.cpp_shift == 30 or p.KH != 7 or p.span_pad != 32 or p.cout != 64 or p.n_pad != 64:
        raise ValueError("stem_pool: needs the fp32 NHWC4 image and the 7x7 / 64-filter row-span stem packing")
    if CONV_MATH not in ("f32", "f32x3"):
        raise ValueError(f"stem_pool: no fused stem in conv math {CONV_MATH!r}")
    x3 = CONV_MATH == "f32x3"
    Hc, Wc = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    Hp, Wp = (Hc + 2 - 3) // 2 + 1, (Wc + 2 - 3) // 2 + 1
    out = torch.empty((B, Hp, Wp, 64), dtype=torch.float32, device=x4.device)
    with _Prof("stem7x7s2_pool_x3" if x3 else "stem7x7s2_pool", 2.0 * B * Hc * Wc * 64 * 147,
               16 * B * H * W + 4 * out.numel() + 4 * 64 * 224, f"B={B} HxW={H}x{W} -> {Hp}x{Wp}x64"):
        fn, w = (lib.ml_stem7x7s2_pool_x3, dc.wgt_x3) if x3 else (lib.ml_stem7x7s2_pool_f32, dc.wgt)
        _lib.check(fn(_ptr(x4), _ptr(w), _ptr(dc.bias), _ptr(out), B, H, W, Hp, Wp, _stream()), "ml_stem7x7s2_pool")
    return out


def maxpool3x3s2(x, pad=1):
    lib = _lib.load()
    _require_dev(x, "x")
    B, H, W, Cc = x.shape
    Ho = (H + 2 * pad - 3) // 2 + 1
    Wo = (W + 2 * pad - 3) // 2 + 1
    half = x.dtype == torch.float16
    out = torch.empty((B, Ho, Wo, Cc), dtype=x.dtype, device=x.device)
    fn = lib.ml_maxpool3x3s2_f16 if half else lib.ml_maxpool3x3s2_f32
    with _Prof("maxpool3x3s2_h" if half else "maxpool3x3s2", 0, x.element_size() * (x.numel() + out.numel())):
        _lib.check(fn(_ptr(x), _ptr(out), B, H, W, Cc, Ho, Wo, pad, pad, _stream()), "ml_maxpool3x3s2")
    return out


def subsample2_h(x):
    """ml_subsample2_f16: x[:, ::2, ::2, :] of a float16 NHWC tensor (what a 1x1 stride-2 conv reads)."""
    lib = _lib.load()
    _require_dev(x, "x")
    if x.dtype != torch.float16:
        raise RuntimeError("subsample2_h: float16 tensor expected")
    B, H, W, Cc = x.shape
    out = torch.empty((B, (H + 1) // 2, (W + 1) // 2, Cc), dtype=torch.float16, device=x.device)
    with _Prof("subsample2_h", 0, 2 * 2 * out.numel()):
        _lib.check(lib.ml_subsample2_f16(_ptr(x), _ptr(out), B, H, W, Cc, _stream()), "ml_subsample2_f16")
    return out


def cast_h2f(x):
    """ml_cast_f16_to_f32: a float16 tensor as float32 (the backbone taps handed to the fp32 heads)."""
    lib = _lib.load()
    _require_dev(x, "x")
    if x.dtype != torch.float16:
        raise RuntimeError("cast_h2f: float16 tensor expected")
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    with _Prof("cast_h2f", 0, 6 * x.numel()):
        _lib.check(lib.ml_cast_f16_to_f32(_ptr(x), _ptr(out), x.numel(), _stream()), "ml_cast_f16_to_f32")
    return out


def cast_f2h(x):
    """ml_cast_f32_to_f16: a float32 tensor as float16 (an fp32 tensor entering an fp16-storage part)."""
    lib = _lib.load()
    _require_dev(x, "x")
    if x.dtype != torch.float32:
        raise RuntimeError("cast_f2h: float32 tensor expected")
    out = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    with _Prof("cast_f2h", 0, 6 * x.numel()):
        _lib.check(lib.ml_cast_f32_to_f16(_ptr(x), _ptr(out), x.numel(), _stream()), "ml_cast_f32_to_f16")
    return out


def preprocess(images, flip, mean, divisor, shift, out_channels=4):
    """BackBonePreProcess fused with the NHWC4 repack.  images: uint8 or float32 [B,H,W,3]."""
    lib = _lib.load()
    _require_dev(images, "images")
    if images.dtype not in (torch.uint8, torch.float32):
        raise TypeError("images must be uint8 or float32")
    B, H, W, ch = images.shape
    if ch != 3:
        raise ValueError("images must have 3 channels (RGB, 0..255)")
    out = torch.empty((B, H, W, out_channels), dtype=torch.float32, device=images.device)
    f3 = C.c_float * 3
    div3 = [float(divisor)] * 3 if np.isscalar(divisor) else [float(v) for v in divisor]
    sh3 = [float(shift)] * 3 if np.isscalar(shift) else [float(v) for v in shift]
    with _Prof("preprocess", 0, images.numel() * images.element_size() + 4 * out.numel()):
        _lib.check(lib.ml_preprocess_f32(_ptr(images), int(images.dtype == torch.uint8), _ptr(out), B * H * W,
                                         out_channels, int(flip), f3(*[float(m) for m in mean]), f3(*div3), f3(*sh3),
                                         _stream()), "ml_preprocess_f32")
    return out


def groupnorm_chunk(x, gamma, beta, groups, eps=1e-5, relu=False, out=None, out_coff=0):
    """`out` may be x itself (in place), a same-shape tensor, or a wider concat buffer
    [N,H,W,Cbuf] written at channels [out_coff, out_coff+C)."""
    lib = _lib.load()
    _require_dev(x, "x")
    N = x.shape[0]
    Cc = x.shape[-1]
    hwc = x.numel() // N
    if out is None:
        out = torch.empty_like(x)
    _require_dev(out, "out")
    out_cs = out.shape[-1]
    if out_cs == Cc:
        if out.numel() != x.numel() or out_coff != 0:
            raise ValueError("groupnorm: dense output must match the input size")
    elif tuple(out.shape[:-1]) != tuple(x.shape[:-1]):
        raise ValueError("groupnorm: concat buffer spatial shape mismatch")
    if out.dtype != x.dtype or x.dtype not in (torch.float32, torch.float16):
        raise ValueError("groupnorm: x and out must share a dtype (float32 or float16)")
    half = x.dtype == torch.float16
    ws = workspace(lib.ml_groupnorm_workspace_bytes(N, groups), x.device, "gn")
    fn = lib.ml_groupnorm_chunk_f16 if half else lib.ml_groupnorm_chunk_f32
    with _Prof("groupnorm_chunk_h" if half else "groupnorm_chunk", 0, 2 * x.element_size() * x.numel(),
               f"N={N} HWC={hwc} G={groups}"):
        _lib.check(fn(_ptr(x), _ptr(out), _ptr(gamma), _ptr(beta), N, hwc, Cc, groups,
                      float(eps), int(relu), out_cs, out_coff, _ptr(ws), _stream()), "ml_groupnorm_chunk")
    return out


def groupnorm_chunk_multi(problems):
    """ml_groupnorm_multi_f32: several independent GroupNormalizations in one launch pair.
    problems: list of dicts (x, gamma, beta, groups, eps, relu=False, out=None, out_coff=0) -> list of outputs."""
    lib = _lib.load()
    n = len(problems)
    if n == 0:
        return []
    if n > _lib.GN_MAX_PROBLEMS:
        return groupnorm_chunk_multi(problems[:_lib.GN_MAX_PROBLEMS]) + groupnorm_chunk_multi(problems[_lib.GN_MAX_PROBLEMS:])
    arr = (_lib.GnDesc * n)()
    outs, nbytes, ws_bytes = [], 0, 0
    for i, pr in enumerate(problems):
        x = pr["x"]
        _require_dev(x, "x")
        out = pr.get("out")
        if out is None:
            out = torch.empty_like(x)
        _require_dev(out, "out")
        N, Cc = x.shape[0], x.shape[-1]
        out_cs, out_coff = out.shape[-1], pr.get("out_coff", 0)
        if out_cs == Cc:
            if out.numel() != x.numel() or out_coff != 0:
                raise ValueError("groupnorm: dense output must match the input size")
        elif tuple(out.shape[:-1]) != tuple(x.shape[:-1]):
            raise ValueError("groupnorm: concat buffer spatial shape mismatch")
        d = arr[i]
        d.x, d.y = x.data_ptr(), out.data_ptr()
        d.gamma = pr["gamma"].data_ptr() if pr.get("gamma") is not None else None
        d.beta = pr["beta"].data_ptr() if pr.get("beta") is not None else None
        d.HWC, d.N, d.C, d.G = x.numel() // N, N, Cc, pr["groups"]
        if out.dtype != x.dtype or x.dtype != problems[0]["x"].dtype:
            raise ValueError("groupnorm_multi: every x / out of one launch must share the dtype")
        d.relu, d.out_cstride, d.out_coff, d.eps = int(pr.get("relu", False)), out_cs, out_coff, float(pr.get("eps", 1e-5))
        d.dtype = int(x.dtype == torch.float16)
        if pr.get("live") is not None:
            lv, period = pr["live"]
            d.live, d.live_period = lv.data_ptr(), int(period)
        if pr.get("partials") is not None:         # (float64 [chunks * n, 2] written by the producing conv, n per chunk)
            pt, npc = pr["partials"]
            d.partials, d.n_partials = pt.data_ptr(), int(npc)
            nbytes -= x.element_size() * x.numel() // 3       # (algorithmic bytes stay 1R + 1W; the pass that is gone was the 3rd)
        ws_bytes += (int(lib.ml_groupnorm_workspace_bytes(N, pr["groups"])) + 255) // 256 * 256
        nbytes += 2 * x.element_size() * x.numel()
        outs.append(out)
    ws = workspace(ws_bytes, problems[0]["x"].device, "gn_multi")
    with _Prof("groupnorm_chunk_h" if problems[0]["x"].dtype == torch.float16 else "groupnorm_chunk", 0, nbytes, f"multi x{n}"):
        _lib.check(lib.ml_groupnorm_multi_f32(arr, n, _ptr(ws), ws.numel(), _stream()), "ml_groupnorm_multi_f32")
    return outs


def resize_bilinear_ac(x, oh, ow, add=None, out=None, out_coff=0):
    lib = _lib.load()
    _require_dev(x, "x")
    B, H, W, Cc = x.shape
    half = x.dtype == torch.float16
    if Cc % 4 != 0:                    # e.g. the 3-class semantic map: scalar kernel, no add / concat view
        if add is not None or out is not None or half:
            raise RuntimeError("resize_bilinear_ac: add=/out=/float16 need a channel count that is a multiple of 4 (8)")
        return resize_image_ac(x, oh, ow)
    if out is None:
        out = torch.empty((B, oh, ow, Cc), dtype=x.dtype, device=x.device)
        out_coff = 0
    if out.dtype != x.dtype or (add is not None and add.dtype != x.dtype):
        raise ValueError("resize_bilinear_ac: x, add and out must share a dtype")
    add_cs = add.shape[3] if add is not None else 0
    fn = lib.ml_resize_bilinear_ac_f16 if half else lib.ml_resize_bilinear_ac_f32
    with _Prof("resize_bilinear_h" if half else "resize_bilinear", 0,
               x.element_size() * (x.numel() + B * oh * ow * Cc * (2 if add is not None else 1))):
        _lib.check(fn(_ptr(x), _ptr(add), _ptr(out), B, H, W, Cc, Cc, 0, oh, ow,
                      add_cs, 0, out.shape[3], out_coff, _stream()), "ml_resize_bilinear_ac")
    return out


def global_mean(x):
    lib = _lib.load()
    _require_dev(x, "x")
    B, H, W, Cc = x.shape
    out = torch.empty((B, 1, 1, Cc), dtype=x.dtype, device=x.device)
    fn = lib.ml_global_mean_f16 if x.dtype == torch.float16 else lib.ml_global_mean_f32
    _lib.check(fn(_ptr(x), _ptr(out), B, H * W, Cc, _stream()), "ml_global_mean")
    return out


def scale_channels_(x, s):
    lib = _lib.load()
    if x.dtype != torch.float32 or s.dtype != torch.float32:
        raise NotImplementedError("scale_channels_: float32 tensors only (SqueezeExcite is not built for fp16 storage)")
    B, H, W, Cc = x.shape
    _lib.check(lib.ml_scale_channels_f32(_ptr(x), _ptr(s), B, H * W, Cc, _stream()), "ml_scale_channels_f32")
    return x


def restore_boxes(loc_pred, priors_i32):
    lib = _lib.load()
    _require_dev(loc_pred, "loc_pred")
    B, A, _ = loc_pred.shape
    boxes = torch.empty((B, A, 4), dtype=torch.float32, device=loc_pred.device)
    _lib.check(lib.ml_restore_boxes_f32(_ptr(loc_pred), _ptr(priors_i32), _ptr(boxes), B, A, _stream()),
               "ml_restore_boxes_f32")
    return boxes


def detection_proposal(cls_pred, boxes, min_confidence, nms_iou, post_iou, max_out, want_kept=False,
                       want_payload=False):
    """Fixed-capacity DetectionProposal: -> proposed [B,max_out,6] (-1 padded), counts [B] int32
    (device), kept [B,max_out,2] int32 or None[, payload [B,max_out*6+1] -- the all-gather record the
    kernel writes beside `proposed`, see parallel.all_gather_detections]."""
    lib = _lib.load()
    _require_dev(cls_pred, "cls_pred")
    _require_dev(boxes, "boxes")
    B, A, Cn = cls_pred.shape
    dev = cls_pred.device
    proposed = torch.empty((B, max_out, 6), dtype=torch.float32, device=dev)
    counts = torch.empty((B,), dtype=torch.int32, device=dev)
    kept = torch.empty((B, max_out, 2), dtype=torch.int32, device=dev) if want_kept else None
    payload = torch.empty((B, max_out * 6 + 1), dtype=torch.float32, device=dev) if want_payload else None
    ws = workspace(lib.ml_detection_workspace_bytes(B, A, Cn, max_out), dev, "det")
    with _Prof("detection_proposal", 0, 4 * (cls_pred.numel() + boxes.numel())):
        _lib.check(lib.ml_detection_proposal_f32(_ptr(cls_pred), _ptr(boxes), _ptr(proposed), _ptr(counts), _ptr(kept),
                                                 _ptr(payload), B, A, Cn, float(min_confidence), float(nms_iou),
                                                 float(post_iou), int(max_out), _ptr(ws), _stream()),
                   "ml_detection_proposal_f32")
    if want_payload:
        return proposed, counts, kept, payload
    return proposed, counts, kept


def mask_distribute(rows, max_k, base_size, has_k=False, want_k=False):
    """rows [B,cap,6] (has_k=False: k computed per MaskDistribute) or [B,cap,7] dist_boxes
    (has_k=True).  -> level_slots [B,L,cap] int32, level_counts [B,L] int32, level_max [L] int32 (max over
    the images: the one thing the host reads), kvals [B,cap] or None."""
    lib = _lib.load()
    _require_dev(rows, "rows")
    B, cap, rs = rows.shape
    L = max_k + 1
    slots = torch.empty((B, L, cap), dtype=torch.int32, device=rows.device)
    lcounts = torch.empty((B, L), dtype=torch.int32, device=rows.device)
    lmax = torch.empty((L,), dtype=torch.int32, device=rows.device)
    kvals = torch.empty((B, cap), dtype=torch.float32, device=rows.device) if want_k else None
    _lib.check(lib.ml_mask_distribute_i32(_ptr(rows), rs, int(has_k), _ptr(kvals), _ptr(slots), _ptr(lcounts), _ptr(lmax),
                                          B, cap, max_k, float(base_size), _stream()), "ml_mask_distribute_i32")
    return slots, lcounts, lmax, kvals


def roi_crop_resize(fmap, rows, slots, lcounts, level, n_l, crop_size, img_hw, roi_boxes, box_off, live=None):
    """rows [B,cap,6] (cx,cy,w,h,cls,conf) or [B,cap,7] dist_boxes (k first).  live: device int32 [1] = the level's
    RoI maximum when the launch runs at capacity (n_l = cap): slots past max(1, live) are not written."""
    lib = _lib.load()
    _require_dev(fmap, "fmap")
    B, Hf, Wf, Cc = fmap.shape
    cap, rs = rows.shape[1], rows.shape[2]
    roff = rs - 6
    L = slots.shape[1]
    ch, cw = crop_size
    out = torch.empty((B, n_l, ch, cw, Cc), dtype=fmap.dtype, device=fmap.device)
    fn = lib.ml_roi_crop_resize_f16 if fmap.dtype == torch.float16 else lib.ml_roi_crop_resize_f32
    _lib.check(fn(_ptr(fmap), _ptr(rows), rs, roff, _ptr(slots), _ptr(lcounts), _ptr(out),
                  _ptr(roi_boxes), B, Hf, Wf, Cc, cap, L, level, n_l, ch, cw,
                  float(img_hw[0]), float(img_hw[1]), box_off, roi_boxes.shape[1], _ptr(live), _stream()), "ml_roi_crop_resize")
    return out


def mold_levels(src, n_l, cap):
    """ml_mold_levels_f32: src [B, L*cap, ...] (level l's RoIs at rows l*cap ..) -> [B, sum(n_l), ...] with the first n_l[l]
    rows of every level next to each other; n_l: host ints."""
    lib = _lib.load()
    _require_dev(src, "src")
    if src.dtype != torch.float32:
        raise RuntimeError("mold_levels: float32 tensor expected")
    B, L = src.shape[0], len(n_l)
    if src.shape[1] != L * cap:
        raise ValueError(f"mold_levels: {src.shape[1]} rows per image, expected {L} x {cap}")
    E = int(np.prod(src.shape[2:]))
    if E % 4:                                      # [B, L*cap, 6] boxes: tiny, plain slicing (data movement only)
        return torch.cat([src[:, l * cap:l * cap + n] for l, n in enumerate(n_l)], dim=1).contiguous()
    out = torch.empty((B, int(sum(n_l))) + tuple(src.shape[2:]), dtype=torch.float32, device=src.device)
    arr = (C.c_int32 * L)(*[int(n) for n in n_l])
    _lib.check(lib.ml_mold_levels_f32(_ptr(src), _ptr(out), B, L, int(cap), E, arr, _stream()), "ml_mold_levels_f32")
    return out


def mold_levels_dev(src, lmax, cap):
    """ml_mold_levels_dev_f32: the same concatenation with the level sizes read on the DEVICE (`lmax`: int32 [L], the
    per-level RoI maxima) -- part of the captured forward.  -> a capacity buffer shaped like `src` whose flat FRONT holds the
    [B, sum n_l, ...] tensor; `molded_front(buf, n_l)` takes it as a view once the host knows n_l."""
    lib = _lib.load()
    _require_dev(src, "src")
    _require_dev(lmax, "lmax")
    if src.dtype != torch.float32 or lmax.dtype != torch.int32:
        raise RuntimeError("mold_levels_dev: float32 tensor and int32 level maxima expected")
    B, L = src.shape[0], int(lmax.numel())
    if src.shape[1] != L * cap:
        raise ValueError(f"mold_levels_dev: {src.shape[1]} rows per image, expected {L} x {cap}")
    E = int(np.prod(src.shape[2:]))
    out = torch.empty_like(src)
    _lib.check(lib.ml_mold_levels_dev_f32(_ptr(src), _ptr(out), B, L, int(cap), E, _ptr(lmax), _stream()), "ml_mold_levels_dev_f32")
    return out


def molded_front(buf, n_l):
    """The [B, sum(n_l), ...] tensor at the front of a mold_levels_dev() buffer (a view: no launch)."""
    B, total = buf.shape[0], int(sum(n_l))
    E = int(np.prod(buf.shape[2:]))
    return buf.view(-1)[:B * total * E].view((B, total) + tuple(buf.shape[2:]))


def add_(x, y):
    """x += y (same shape)."""
    lib = _lib.load()
    _require_dev(x, "x")
    _require_dev(y, "y")
    if x.shape != y.shape:
        raise ValueError("add_: shape mismatch")
    if x.dtype != torch.float32 or y.dtype != torch.float32:
        raise NotImplementedError("add_: float32 tensors only (MobileSeparableConv2D is not built for fp16 storage)")
    _lib.check(lib.ml_add_f32(_ptr(x), _ptr(y), x.numel(), _stream()), "ml_add_f32")
    return x


def fill_(x, v):
    lib = _lib.load()
    _lib.check(lib.ml_fill_f32(_ptr(x), float(v), x.numel(), _stream()), "ml_fill_f32")
    return x


# ----------------------------------------------------------------------------- deploy wrapper
def resize_image_ac(x, oh, ow, threshold=None):
    """ml_resize_image_ac: bilinear(align_corners=True) of a [B,H,W,C] uint8 / float32 tensor, any C.
    threshold=None -> float32 result; else int32 `value > threshold`."""
    lib = _lib.load()
    _require_dev(x, "x")
    if x.dtype not in (torch.uint8, torch.float32):
        raise RuntimeError(f"resize_image_ac: uint8 or float32 input, got {x.dtype}")
    B, H, W, Cc = x.shape
    out_f = out_i = None
    if threshold is None:
        out_f = torch.empty((B, oh, ow, Cc), dtype=torch.float32, device=x.device)
    else:
        out_i = torch.empty((B, oh, ow, Cc), dtype=torch.int32, device=x.device)
    with _Prof("resize_image", 0, x.numel() * x.element_size() + 4 * B * oh * ow * Cc):
        _lib.check(lib.ml_resize_image_ac(_ptr(x), int(x.dtype == torch.uint8), _ptr(out_f), _ptr(out_i),
                                          float(threshold if threshold is not None else 0.0), B, H, W, Cc, oh, ow,
                                          _stream()), "ml_resize_image_ac")
    return out_f if threshold is None else out_i


def trim_instances(roi_boxes, roi_masks):
    """ml_trim_instances_f32 -> (boxes [B,N,6], masks [B,N,mh,mw], counts [B] int32), fixed capacity."""
    lib = _lib.load()
    _require_dev(roi_boxes, "roi_boxes")
    _require_dev(roi_masks, "roi_masks")
    B, N, six = roi_boxes.shape
    if six != 6 or roi_masks.dim() != 5 or tuple(roi_masks.shape[:2]) != (B, N):
        raise ValueError(f"trim_instances: roi_boxes [B,N,6] / roi_masks [B,N,h,w,C] expected, got "
                         f"{tuple(roi_boxes.shape)} / {tuple(roi_masks.shape)}")
    _, _, mh, mw, Cc = roi_masks.shape
    out_b = torch.empty((B, N, 6), dtype=torch.float32, device=roi_boxes.device)
    out_m = torch.empty((B, N, mh, mw), dtype=torch.float32, device=roi_boxes.device)
    counts = torch.empty((B,), dtype=torch.int32, device=roi_boxes.device)
    with _Prof("trim_instances", 0, 4 * (roi_boxes.numel() * 2 + out_m.numel() * 2)):
        _lib.check(lib.ml_trim_instances_f32(_ptr(roi_boxes), _ptr(roi_masks), _ptr(out_b), _ptr(out_m), _ptr(counts),
                                             B, N, mh, mw, Cc, _stream()), "ml_trim_instances_f32")
    return out_b, out_m, counts


def upsample_boxes(rows, ratio0, ratio1):
    lib = _lib.load()
    _require_dev(rows, "rows")
    out = torch.empty(rows.shape, dtype=torch.int32, device=rows.device)
    _lib.check(lib.ml_upsample_boxes_i32(_ptr(rows), _ptr(out), rows.numel() // 6, float(ratio0), float(ratio1),
                                         _stream()), "ml_upsample_boxes_i32")
    return out


def threshold_i32(x, threshold=0.5):
    lib = _lib.load()
    _require_dev(x, "x")
    out = torch.empty(x.shape, dtype=torch.int32, device=x.device)
    _lib.check(lib.ml_threshold_i32(_ptr(x), _ptr(out), float(threshold), x.numel(), _stream()), "ml_threshold_i32")
    return out


def semantic_smoothing(x, kernel_sizes, weights):
    """ml_semantic_smoothing_f32: per-class grey opening (erosion -> dilation, flat k x k) times weight."""
    lib = _lib.load()
    _require_dev(x, "x")
    B, H, W, Cc = x.shape
    if len(kernel_sizes) != Cc or len(weights) != Cc:
        raise ValueError(f"semantic_smoothing: {Cc} classes need {Cc} kernel sizes and weights")
    ks = (C.c_int32 * Cc)(*[int(k) for k in kernel_sizes])
    ws = (C.c_float * Cc)(*[float(w) for w in weights])
    out = torch.empty_like(x)
    tmp = torch.empty_like(x)
    with _Prof("semantic_smoothing", 0, 4 * x.numel() * 8):
        _lib.check(lib.ml_semantic_smoothing_f32(_ptr(x), _ptr(out), _ptr(tmp), B, H, W, Cc, ks, ws, _stream()),
                   "ml_semantic_smoothing_f32")
    return out


# ----------------------------------------------------------------------------- serving post-processing
def instance_summary_rois(seg, det_outs, ins_outs, road_channel=1, default_road_size=3.25, ioi_threshold=0.1):
    """ml_instance_summary_rois_f32: crop_pad_mask + instance_summary without the [B,n,H,W] canvases (bit-identical).
    seg int32 [B,H,W,C], det_outs int32 [B,n,6], ins_outs int32 [B,n,mh,mw] -> [B,n,5]."""
    lib = _lib.load()
    for t, name in ((seg, "seg"), (det_outs, "det_outs"), (ins_outs, "ins_outs")):
        _require_dev(t, name)
        if t.dtype != torch.int32:
            raise RuntimeError("instance_summary_rois: int32 semantic map, detections and masks expected")
    B, H, W, Cc = seg.shape
    _, n, mh, mw = ins_outs.shape
    out = torch.empty((B, n, 5), dtype=torch.float32, device=seg.device)
    ws = workspace(int(lib.ml_instance_summary_workspace_bytes(B, H)), seg.device, "summary")
    with _Prof("instance_summary_rois", 0, 4 * (ins_outs.numel() + 2 * seg.numel())):
        _lib.check(lib.ml_instance_summary_rois_f32(_ptr(seg), Cc, int(road_channel), _ptr(det_outs), _ptr(ins_outs), _ptr(out),
                                                    B, n, mh, mw, H, W, float(default_road_size), float(ioi_threshold),
                                                    _ptr(ws), _stream()), "ml_instance_summary_rois_f32")
    return out


def crop_pad_mask(det_outs, ins_outs, height, width):
    """ml_crop_pad_mask_f32: det [B,n,6] int32, masks [B,n,mh,mw] int32 -> [B,n,H,W] float32."""
    lib = _lib.load()
    _require_dev(det_outs, "det_outs")
    _require_dev(ins_outs, "ins_outs")
    if det_outs.dtype != torch.int32 or ins_outs.dtype != torch.int32:
        raise RuntimeError("crop_pad_mask: int32 detections and masks expected (UpSampleOutput's outputs)")
    B, n, _ = det_outs.shape
    _, _, mh, mw = ins_outs.shape
    out = torch.empty((B, n, height, width), dtype=torch.float32, device=det_outs.device)
    thr = torch.empty((1,), dtype=torch.int32, device=det_outs.device)
    with _Prof("crop_pad_mask", 0, 4 * (out.numel() + ins_outs.numel())):
        _lib.check(lib.ml_crop_pad_mask_f32(_ptr(det_outs), _ptr(ins_outs), _ptr(out), _ptr(thr), B, n, mh, mw,
                                            int(height), int(width), _stream()), "ml_crop_pad_mask_f32")
    return out


def nonzero_bbox(seg, channel):
    """ml_nonzero_bbox_i32 -> int32 [5] = (ymin, xmin, ymax, xmax, any) over the whole batch."""
    lib = _lib.load()
    _require_dev(seg, "seg")
    B, H, W, Cc = seg.shape
    box = torch.tensor([2 ** 31 - 1, 2 ** 31 - 1, -1, -1, 0], dtype=torch.int32, device=seg.device)
    _lib.check(lib.ml_nonzero_bbox_i32(_ptr(seg), B, H, W, Cc, int(channel), _ptr(box), _stream()), "ml_nonzero_bbox_i32")
    return box


def instance_summary(seg, masks, road_channel=1, default_road_size=3.25, ioi_threshold=0.1):
    """ml_instance_summary_f32 -> [B,n,5] = (pixel sum, instance size, horizontal, vertical, include_my_road)."""
    lib = _lib.load()
    _require_dev(seg, "seg")
    _require_dev(masks, "masks")
    if seg.dtype != torch.int32 or masks.dtype != torch.float32:
        raise RuntimeError("instance_summary: int32 semantic map and float32 padded masks expected")
    B, H, W, Cc = seg.shape
    _, n, _, _ = masks.shape
    out = torch.empty((B, n, 5), dtype=torch.float32, device=seg.device)
    ws = workspace(int(lib.ml_instance_summary_workspace_bytes(B, H)), seg.device, "summary")
    with _Prof("instance_summary", 0, 4 * (2 * masks.numel() + 2 * seg.numel())):
        _lib.check(lib.ml_instance_summary_f32(_ptr(seg), Cc, int(road_channel), _ptr(masks), _ptr(out), B, n, H, W,
                                               float(default_road_size), float(ioi_threshold), _ptr(ws), _stream()),
                   "ml_instance_summary_f32")
    return out
