"""Host-side weight re-layout (done once at load time) and convolution geometry.

Keras layouts in -> the layouts the gfx950 kernels read:
  * dense conv   kernel[kh,kw,cin,cout]          -> wgt[n_pad][taps*span_pad], k = tap*span_pad + c
  * stem / 3-channel convs on the NHWC4 image    -> one tap per kernel ROW, span = KW*4 floats
  * ResNeXt grouped 3x3 (reference spells it DepthwiseConv2D(depth_multiplier=c) + reshape +
    reduce_sum, engine/backbone/ResNext.py:212-219) -> block-diagonal dense rows, 32-wide N tiles
  * Conv2DTranspose(2,2,s2) kernel[2,2,cout,cin] -> 1x1 GEMM with N = 4*cout, pixel-shuffled
  * DepthwiseConv2D kernel[3,3,C,1]              -> [9][C]
  * BatchNormalization (inference) folded into the preceding conv's weights / bias.
Pure numpy; no GPU needed (unit-tested on CPU against the oracle through tests/emulate.py).
"""
import math
from dataclasses import dataclass, field

import numpy as np

NO_PIX_SPAN = 30  # cpp_shift value meaning "a tap never spans pixels"


def same_pads(size, k, stride, dilation=1):
    """TF padding='same': out = ceil(in/stride); the odd pixel goes after."""
    out = -(-size // stride)
    k_eff = (k - 1) * dilation + 1
    total = max((out - 1) * stride + k_eff - size, 0)
    return out, total // 2, total - total // 2


def resolve_padding(H, W, kh, kw, stride, dilation, padding):
    """-> (Ho, Wo, pad_top, pad_left) for padding in {'same','valid',((t,b),(l,r))}."""
    if padding == "same":
        Ho, pt, _ = same_pads(H, kh, stride, dilation)
        Wo, pl, _ = same_pads(W, kw, stride, dilation)
        return Ho, Wo, pt, pl
    if padding == "valid":
        pt = pb = pl = pr = 0
    else:
        (pt, pb), (pl, pr) = padding
    Ho = (H + pt + pb - ((kh - 1) * dilation + 1)) // stride + 1
    Wo = (W + pl + pr - ((kw - 1) * dilation + 1)) // stride + 1
    return Ho, Wo, pt, pl


def ntile_for(cout, tile=0):
    """N-tile width of the MFMA conv kernel's auto heuristic (mirrors ml_conv2d_ntile)."""
    if tile in (1, 2, 3, 4, 5):       # 4 / 5 = the pipelined 1x1 kernels (packed in 128-wide tiles like 1)
        return {1: 128, 2: 64, 3: 32, 4: 128, 5: 128}[tile]
    if cout <= 32:
        return 32
    if cout <= 64:
        return 64
    if cout <= 96:
        return 32
    return 128


@dataclass
class PackedConv:
    """Everything ml_conv2d_f32 needs except tensor pointers and spatial sizes."""
    wgt: np.ndarray                 # [n_pad, ktot] float32
    bias: np.ndarray = None         # [cout_per_pixel] float32 or None
    KH: int = 1
    KW: int = 1
    span: int = 0
    span_pad: int = 0
    cpp_shift: int = NO_PIX_SPAN
    cout: int = 0                   # N (4*Cout for the transposed conv)
    n_pad: int = 0
    group_cin_step: int = 0
    shuffle2x2: int = 0
    tile: int = 0
    cin_buffer: int = 0             # channels the input buffer must have (4 for NHWC4 inputs)
    kh_real: int = 1                # real kernel geometry (for padding resolution)
    kw_real: int = 1
    k_real: int = 0                 # algorithmic MACs per output element (for roofline accounting)
    extra: dict = field(default_factory=dict)


def _pad_rows(w2d, cout, tile):
    bn = ntile_for(cout, tile)
    n_pad = -(-cout // bn) * bn
    out = np.zeros((n_pad, w2d.shape[1]), np.float32)
    out[:cout] = w2d
    return out, n_pad


def fold_bn(kernel, bias, gamma, beta, mean, var, eps, depthwise=False):
    """Fold inference BatchNormalization into the preceding (bias-free or biased) conv."""
    scale = (1.0 if gamma is None else gamma.astype(np.float64)) / np.sqrt(var.astype(np.float64) + eps)
    if depthwise:                       # kernel [kh,kw,C,1]
        k = kernel.astype(np.float64) * scale[None, None, :, None]
    else:                               # kernel [kh,kw,cin,cout]
        k = kernel.astype(np.float64) * scale[None, None, None, :]
    b0 = 0.0 if bias is None else bias.astype(np.float64)
    b = (b0 - mean.astype(np.float64)) * scale + beta.astype(np.float64)
    return k.astype(np.float32), b.astype(np.float32)


def pack_dense(kernel, bias=None, tile=0):
    """Conv2D kernel [kh,kw,cin,cout] with cin % 4 == 0."""
    kh, kw, cin, cout = kernel.shape
    if cin % 4:
        raise ValueError(f"dense conv needs cin % 4 == 0 (got {cin}); use pack_rowspan for image inputs")
    span_pad = -(-cin // 32) * 32
    w = np.zeros((cout, kh, kw, span_pad), np.float32)
    w[..., :cin] = np.transpose(kernel, (3, 0, 1, 2))
    w2d, n_pad = _pad_rows(w.reshape(cout, kh * kw * span_pad), cout, tile)
    return PackedConv(wgt=w2d, bias=None if bias is None else np.ascontiguousarray(bias, np.float32),
                      KH=kh, KW=kw, span=cin, span_pad=span_pad, cout=cout, n_pad=n_pad, tile=tile,
                      cin_buffer=cin, kh_real=kh, kw_real=kw, k_real=kh * kw * cin)


def pack_rowspan(kernel, bias=None, cpad=4, tile=0):
    """Conv on a channel-padded image (NHWC4): one MFMA tap per kernel ROW; the tap's K span is the
    KW adjacent pixels x cpad floats, contiguous in memory.  Used for the 3-channel stems."""
    kh, kw, cin, cout = kernel.shape
    assert cin <= cpad and cpad in (4, 8, 16)
    span = kw * cpad
    span_pad = -(-span // 32) * 32
    w = np.zeros((cout, kh, span_pad), np.float32)
    kt = np.transpose(kernel, (3, 0, 1, 2))            # [cout,kh,kw,cin]
    for j in range(kw):
        w[:, :, j * cpad:j * cpad + cin] = kt[:, :, j, :]
    w2d, n_pad = _pad_rows(w.reshape(cout, kh * span_pad), cout, tile)
    return PackedConv(wgt=w2d, bias=None if bias is None else np.ascontiguousarray(bias, np.float32),
                      KH=kh, KW=1, span=span, span_pad=span_pad, cpp_shift=int(math.log2(cpad)),
                      cout=cout, n_pad=n_pad, tile=tile, cin_buffer=cpad, kh_real=kh, kw_real=kw,
                      k_real=kh * kw * cin)


def pack_grouped(dw_kernel, groups, bias=None):
    """ResNeXt grouped 3x3.  Reference weight: DepthwiseConv2D kernel K[kh,kw,in,m] with
    in = g*c+i, depth_multiplier = c; out channel g*c+m = sum_i conv(x[g*c+i], K[..,g*c+i,m])
    (ResNext.py:212-219).  Packed as dense rows over a 32-channel input window
    per 32-wide N tile (block diagonal inside the tile when c < 32)."""
    kh, kw, filters, c = dw_kernel.shape
    assert filters == groups * c
    if 32 % c and c % 32:
        raise ValueError(f"grouped conv: channels per group {c} must divide 32")
    if c > 32:
        raise NotImplementedError("grouped conv with more than 32 channels per group")
    w = np.zeros((filters, kh, kw, 32), np.float32)
    n = np.arange(filters)
    g = n // c
    m = n % c
    win0 = (n // 32) * 32
    for i in range(c):
        cin_idx = g * c + i
        w[n, :, :, cin_idx - win0] = np.transpose(dw_kernel[:, :, cin_idx, m], (2, 0, 1))
    w2d, n_pad = _pad_rows(w.reshape(filters, kh * kw * 32), filters, 3)
    assert n_pad == filters or filters % 32
    return PackedConv(wgt=w2d, bias=None if bias is None else np.ascontiguousarray(bias, np.float32),
                      KH=kh, KW=kw, span=32, span_pad=32, cout=filters, n_pad=n_pad, group_cin_step=32,
                      tile=3, cin_buffer=filters, kh_real=kh, kw_real=kw, k_real=kh * kw * c)


def pack_grouped_mfma4(dw_kernel, groups):
    """ResNeXt grouped 3x3 for ml_gconv3x3_f32: [C][9][c] with wgt[g*c+m][tap][i] = K[tap][g*c+i][m]."""
    kh, kw, filters, c = dw_kernel.shape
    assert (kh, kw) == (3, 3) and filters == groups * c
    k = dw_kernel.reshape(3, 3, groups, c, c)                 # [kh,kw,g,i,m]
    return np.ascontiguousarray(np.transpose(k, (2, 4, 0, 1, 3)).reshape(filters, 9, c), np.float32)


def grouped_dw_to_dense(dw_kernel, groups):
    """The same re-layout as a plain grouped weight Wg[kh,kw,i,g*c+m] = K[kh,kw,g*c+i,m] (for tests)."""
    kh, kw, filters, c = dw_kernel.shape
    wg = np.zeros((kh, kw, c, filters), dw_kernel.dtype)
    for g in range(groups):
        for i in range(c):
            wg[:, :, i, g * c:(g + 1) * c] = dw_kernel[:, :, g * c + i, :]
    return wg


def pack_transpose2x2(kernel, bias=None, tile=0):
    """Conv2DTranspose(f,(2,2),(2,2)) kernel [2,2,cout,cin]: out[2i+a,2j+b,o] = sum_c in[i,j,c]*K[a,b,o,c].
    -> GEMM with N = 4*cout, column n = (a*2+b)*cout + o, epilogue pixel-shuffles."""
    assert kernel.shape[:2] == (2, 2)
    _, _, cout, cin = kernel.shape
    if cin % 4:
        raise ValueError("transposed conv needs cin % 4 == 0")
    span_pad = -(-cin // 32) * 32
    w = np.zeros((4 * cout, span_pad), np.float32)
    w[:, :cin] = kernel.reshape(4 * cout, cin)
    w2d, n_pad = _pad_rows(w, 4 * cout, tile)
    return PackedConv(wgt=w2d, bias=None if bias is None else np.ascontiguousarray(bias, np.float32),
                      KH=1, KW=1, span=cin, span_pad=span_pad, cout=4 * cout, n_pad=n_pad, shuffle2x2=1,
                      tile=tile, cin_buffer=cin, k_real=cin)


def pack_out1x1_table(kernel, bias=None):
    """1x1 conv kernel [1,1,C_mid,ncls] -> the lane table of ml_deconv2x2_out1x1_f32 (include/masklab_hip.h):
    [C_mid/32][16][2][cp], entry (t, e, half, c) = kernel[32 t + (e & 3) + 8 (e >> 2) + 4 half, c]; cp = the power of two
    >= ncls.  Returns (table, bias padded to cp, cp)."""
    assert kernel.shape[:2] == (1, 1)
    cmid, ncls = int(kernel.shape[2]), int(kernel.shape[3])
    if cmid % 32 or ncls > 32:
        raise ValueError("fused mask-head tail needs C_mid % 32 == 0 and at most 32 classes")
    cp = 1
    while cp < ncls:
        cp *= 2
    t, e, h = np.meshgrid(np.arange(cmid // 32), np.arange(16), np.arange(2), indexing="ij")
    ch = 32 * t + (e & 3) + 8 * (e >> 2) + 4 * h
    table = np.zeros((cmid // 32, 16, 2, cp), np.float32)
    table[..., :ncls] = kernel[0, 0][ch]
    b = np.zeros((cp,), np.float32)
    if bias is not None:
        b[:ncls] = bias
    return np.ascontiguousarray(table), b, cp


def pack_depthwise(dw_kernel):
    """DepthwiseConv2D kernel [3,3,C,1] -> [9][C]."""
    kh, kw, C, mult = dw_kernel.shape
    assert (kh, kw, mult) == (3, 3, 1)
    return np.ascontiguousarray(dw_kernel.reshape(9, C), np.float32)
