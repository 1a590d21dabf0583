"""CPU ORACLE (test infrastructure only) -- TensorFlow-op semantics in NumPy.

PARITY UNPINNED: the reference (craftsangjae/instance-segmentation-road-project)
delegates all arithmetic to TensorFlow 1.14/1.15, which is not installed here and
ships no tests / golden vectors.  These functions restate
the documented behaviour of the TF ops that the hot path calls (SURVEY.md
Appendix A); each cites the reference call site that uses it.  The only
reference-pinned artefact is the anchor table (tests/golden/prior_tables.npz).

Nothing under oracle/ may be imported by the product package; only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.

All tensors are NHWC numpy arrays.  Functions compute in the dtype of `x`
(float32 or float64); weights are cast to it.
"""
import math

import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------- padding
def same_pads(size, k, stride, dilation=1):
    """TF `padding='same'`: out=ceil(in/stride); extra pixel goes AFTER.
    Used by every `padding='same'` conv: reference engine/backbone/base.py:306-312,
    engine/layers/detection.py:42-48,120,127,190,197, semantic.py:63-64."""
    out = -(-size // stride)
    k_eff = (k - 1) * dilation + 1
    total = max((out - 1) * stride + k_eff - size, 0)
    before = total // 2
    return out, before, total - before


def _resolve_padding(x, kh, kw, stride, dilation, padding):
    H, W = x.shape[1], x.shape[2]
    if padding == "same":
        Ho, pt, pb = same_pads(H, kh, stride, dilation)
        Wo, pl, pr = same_pads(W, kw, stride, dilation)
    elif padding == "valid":
        pt = pb = pl = pr = 0
        Ho = (H - ((kh - 1) * dilation + 1)) // stride + 1
        Wo = (W - ((kw - 1) * dilation + 1)) // stride + 1
    else:  # explicit ((top,bottom),(left,right)) then 'valid'  (ZeroPadding2D + valid)
        (pt, pb), (pl, pr) = padding
        Ho = (H + pt + pb - ((kh - 1) * dilation + 1)) // stride + 1
        Wo = (W + pl + pr - ((kw - 1) * dilation + 1)) // stride + 1
    return Ho, Wo, pt, pb, pl, pr


def _pad(x, pt, pb, pl, pr):
    if pt or pb or pl or pr:
        return np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    return x


# --------------------------------------------------------------------------- convolutions
def conv2d(x, w, b=None, stride=1, padding="same", dilation=1):
    """tf.keras Conv2D: cross-correlation, kernel [kh,kw,cin,cout], NHWC, +bias.
    Call sites: ResNext.py:200-226,344; base.py:295-312; detection.py:42-48,120-128,
    190-200; instance.py:188-199; semantic.py:66,112,126,133,199,213,219."""
    kh, kw, cin, cout = w.shape
    assert x.shape[3] == cin, (x.shape, w.shape)
    w = w.astype(x.dtype, copy=False)
    Ho, Wo, pt, pb, pl, pr = _resolve_padding(x, kh, kw, stride, dilation, padding)
    xp = _pad(x, pt, pb, pl, pr)
    B = x.shape[0]
    out = np.zeros((B * Ho * Wo, cout), x.dtype)
    for i in range(kh):
        for j in range(kw):
            y0, x0 = i * dilation, j * dilation
            xs = xp[:, y0:y0 + (Ho - 1) * stride + 1:stride,
                    x0:x0 + (Wo - 1) * stride + 1:stride, :]
            out += xs.reshape(-1, cin) @ w[i, j]
    out = out.reshape(B, Ho, Wo, cout)
    if b is not None:
        out = out + b.astype(x.dtype, copy=False)
    return out


def depthwise_conv2d(x, w, stride=1, padding="same", dilation=1):
    """tf.keras DepthwiseConv2D: kernel [kh,kw,cin,mult]; output channel = cin_idx*mult + m.
    Call sites: ResNext.py:214 (mult=c), semantic.py:63 (dilated), misc.py:85, MobileNet."""
    kh, kw, cin, mult = w.shape
    assert x.shape[3] == cin
    w = w.astype(x.dtype, copy=False)
    Ho, Wo, pt, pb, pl, pr = _resolve_padding(x, kh, kw, stride, dilation, padding)
    xp = _pad(x, pt, pb, pl, pr)
    B = x.shape[0]
    out = np.zeros((B, Ho, Wo, cin, mult), x.dtype)
    for i in range(kh):
        for j in range(kw):
            y0, x0 = i * dilation, j * dilation
            xs = xp[:, y0:y0 + (Ho - 1) * stride + 1:stride,
                    x0:x0 + (Wo - 1) * stride + 1:stride, :]
            out += xs[..., None] * w[i, j]
    return out.reshape(B, Ho, Wo, cin * mult)


def conv2d_transpose_2x2_s2(x, w, b=None):
    """Conv2DTranspose(f,(2,2),(2,2),'same'): kernel [2,2,cout,cin];
    out[2i+a,2j+b,o] = sum_c in[i,j,c]*K[a,b,o,c] + bias[o].  instance.py:195."""
    B, H, W, cin = x.shape
    assert w.shape[:2] == (2, 2) and w.shape[3] == cin
    cout = w.shape[2]
    w = w.astype(x.dtype, copy=False)
    out = np.zeros((B, 2 * H, 2 * W, cout), x.dtype)
    xf = x.reshape(-1, cin)
    for a in range(2):
        for c in range(2):
            out[:, a::2, c::2, :] = (xf @ w[a, c].T).reshape(B, H, W, cout)
    if b is not None:
        out = out + b.astype(x.dtype, copy=False)
    return out


def max_pool(x, k=3, stride=2):
    """MaxPooling2D(3, strides=2) 'valid' (after explicit zero pad) -- ResNext.py:351-352."""
    B, H, W, C = x.shape
    Ho = (H - k) // stride + 1
    Wo = (W - k) // stride + 1
    out = None
    for i in range(k):
        for j in range(k):
            xs = x[:, i:i + (Ho - 1) * stride + 1:stride, j:j + (Wo - 1) * stride + 1:stride, :]
            out = xs.copy() if out is None else np.maximum(out, xs)
    return out


def batch_norm(x, gamma, beta, mean, var, eps):
    """BatchNormalization inference: (x-mean)/sqrt(var+eps)*gamma+beta.
    eps 1.001e-5 ResNext.py:202; 1e-3 MobileNet (Keras default); gamma=None => scale=False."""
    dt = x.dtype
    inv = 1.0 / np.sqrt(var.astype(dt) + dt.type(eps))
    y = (x - mean.astype(dt)) * inv
    if gamma is not None:
        y = y * gamma.astype(dt)
    return y + beta.astype(dt)


def relu(x):
    return np.maximum(x, 0)


def relu6(x):
    return np.minimum(np.maximum(x, 0), 6)


def sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x))).astype(x.dtype)


# --------------------------------------------------------------------------- GroupNormalization
def group_norm(x, gamma, beta, groups, eps=1e-5):
    """LITERAL restatement of reference engine/normalization.py:116-160 (axis=-1):
    reshape [N,H,W,C] -> [N,G,H,W,C/G] ROW-MAJOR (so groups are contiguous chunks of
    the flat H*W*C vector, not channel groups -- SURVEY F5), mean/var over axes
    (2,3,4) (:140-141), normalise (:143), gamma/beta reshaped to [1,G,1,1,C/G]
    (:121-125,151-156), reshape back (:158)."""
    N, H, W, C = x.shape
    if C < groups:
        raise ValueError('Number of groups (' + str(groups) + ') cannot be '
                         'more than the number of channels (' + str(C) + ').')
    if C % groups != 0:
        raise ValueError('Number of groups (' + str(groups) + ') must be a '
                         'multiple of the number of channels (' + str(C) + ').')
    dt = x.dtype
    g = x.reshape(N, groups, H, W, C // groups)
    mean = g.mean(axis=(2, 3, 4), keepdims=True, dtype=dt)
    var = np.mean(np.square(g - mean), axis=(2, 3, 4), keepdims=True, dtype=dt)
    g = (g - mean) / np.sqrt(var + dt.type(eps))
    bshape = (1, groups, 1, 1, C // groups)
    if gamma is not None:
        g = g * gamma.astype(dt).reshape(bshape)
    if beta is not None:
        g = g + beta.astype(dt).reshape(bshape)
    return g.reshape(N, H, W, C)


def group_norm_flat(x, gamma, beta, groups, eps=1e-5):
    """Independent flat-index formulation of the same chunk-norm:
    f=(h*W+w)*C+c ; chunk g=f//(HWC/G) ; param index j=g*(C/G)+(c mod C/G).
    Used only to cross-check group_norm() above."""
    N, H, W, C = x.shape
    L = H * W * C // groups
    flat = x.reshape(N, groups, L).astype(np.float64)
    mean = flat.mean(axis=2, keepdims=True)
    var = flat.var(axis=2, keepdims=True)
    y = (flat - mean) / np.sqrt(var + eps)
    f = np.arange(H * W * C)
    j = (f // L) * (C // groups) + (f % C) % (C // groups)
    y = y.reshape(N, -1) * gamma.astype(np.float64)[j] + beta.astype(np.float64)[j]
    return y.reshape(N, H, W, C)


# --------------------------------------------------------------------------- resampling
def resize_bilinear_align_corners(x, oh, ow):
    """tf.compat.v1.image.resize_bilinear(align_corners=True): scale=(in-1)/(out-1) if
    out>1 else 0; s=o*scale (fp32); lo=floor(s); hi=min(ceil(s),in-1); t=s-lo;
    top=tl+(tr-tl)*tx; bot=bl+(br-bl)*tx; out=top+(bot-top)*ty.
    Call sites: misc.py:306 <- detection.py:58, semantic.py:152,226; misc.py:153,193."""
    B, H, W, C = x.shape
    dt = x.dtype

    def weights(n_in, n_out):
        scale = F32((n_in - 1) / float(n_out - 1)) if n_out > 1 else F32(0.0)
        s = np.arange(n_out, dtype=F32) * scale
        lo = np.maximum(np.floor(s), 0).astype(np.int64)
        hi = np.minimum(np.ceil(s).astype(np.int64), n_in - 1)
        t = (s - np.floor(s)).astype(dt)
        return lo, hi, t

    ylo, yhi, ty = weights(H, oh)
    xlo, xhi, tx = weights(W, ow)
    tx = tx[None, None, :, None]
    ty = ty[None, :, None, None]
    rows_t, rows_b = x[:, ylo], x[:, yhi]
    top = rows_t[:, :, xlo] + (rows_t[:, :, xhi] - rows_t[:, :, xlo]) * tx
    bot = rows_b[:, :, xlo] + (rows_b[:, :, xhi] - rows_b[:, :, xlo]) * tx
    return (top + (bot - top) * ty).astype(dt)


def crop_and_resize(image, boxes, box_ind, crop_size, extrapolation_value=0.0):
    """tf.image.crop_and_resize (bilinear): boxes [n,(y1,x1,y2,x2)] normalised;
    in_y = y1*(H-1) + i*(y2-y1)*(H-1)/(ch-1); outside [0,H-1] => extrapolation value;
    top=floor, bottom=ceil; ONE sample per output cell.  Coordinates are computed in
    float32 exactly as the TF CPU kernel does.  Call site: instance.py:125."""
    B, H, W, C = image.shape
    ch, cw = crop_size
    n = boxes.shape[0]
    dt = image.dtype
    out = np.full((n, ch, cw, C), extrapolation_value, dt)
    boxes = boxes.astype(F32)
    for r in range(n):
        y1, x1, y2, x2 = boxes[r]
        b = int(box_ind[r])
        hs = F32((y2 - y1) * F32(H - 1) / F32(ch - 1)) if ch > 1 else F32(0)
        ws = F32((x2 - x1) * F32(W - 1) / F32(cw - 1)) if cw > 1 else F32(0)
        for i in range(ch):
            in_y = F32(y1 * F32(H - 1) + F32(i) * hs) if ch > 1 else F32(F32(0.5) * (y1 + y2) * F32(H - 1))
            if in_y < 0 or in_y > H - 1:
                continue
            ty_i, by_i = int(math.floor(in_y)), int(math.ceil(in_y))
            ly = dt.type(in_y - F32(ty_i))
            in_x = (x1 * F32(W - 1) + np.arange(cw, dtype=F32) * ws).astype(F32) if cw > 1 else \
                np.full((1,), F32(0.5) * (x1 + x2) * F32(W - 1), F32)
            ok = ~((in_x < 0) | (in_x > W - 1))
            if not ok.any():
                continue
            ixs = np.where(ok, in_x, 0)
            lx_i = np.floor(ixs).astype(np.int64)
            rx_i = np.ceil(ixs).astype(np.int64)
            lx = (ixs - np.floor(ixs)).astype(dt)[:, None]
            tl, tr = image[b, ty_i, lx_i], image[b, ty_i, rx_i]
            bl, br = image[b, by_i, lx_i], image[b, by_i, rx_i]
            top = tl + (tr - tl) * lx
            bot = bl + (br - bl) * lx
            val = top + (bot - top) * ly
            out[r, i, ok] = val[ok]
    return out


# --------------------------------------------------------------------------- NMS & friends
def _iou_f32(bi, bj):
    """TF non_max_suppression_op.cc IOU(), float32 op-for-op."""
    ymin_i, ymax_i = min(bi[0], bi[2]), max(bi[0], bi[2])
    xmin_i, xmax_i = min(bi[1], bi[3]), max(bi[1], bi[3])
    ymin_j, ymax_j = min(bj[0], bj[2]), max(bj[0], bj[2])
    xmin_j, xmax_j = min(bj[1], bj[3]), max(bj[1], bj[3])
    area_i = F32(F32(ymax_i - ymin_i) * F32(xmax_i - xmin_i))
    area_j = F32(F32(ymax_j - ymin_j) * F32(xmax_j - xmin_j))
    if area_i <= 0 or area_j <= 0:
        return F32(0.0)
    iy1, ix1 = max(ymin_i, ymin_j), max(xmin_i, xmin_j)
    iy2, ix2 = min(ymax_i, ymax_j), min(xmax_i, xmax_j)
    inter = F32(max(F32(iy2 - iy1), F32(0.0)) * max(F32(ix2 - ix1), F32(0.0)))
    return F32(inter / F32(F32(area_i + area_j) - inter))


def non_max_suppression(boxes, scores, max_output_size, iou_threshold):
    """tf.image.non_max_suppression: greedy by score descending, keep unless
    IoU(candidate, any kept) > iou_threshold (strict); stop at max_output_size;
    returns indices into the input in selection order.  Equal scores: TF1.x leaves
    the order unspecified; this oracle (and the build) take the LOWER index first.
    Call sites: detection.py:507-510, 542-545."""
    boxes = np.asarray(boxes, F32)
    scores = np.asarray(scores, F32)
    order = np.lexsort((np.arange(len(scores)), -scores.astype(np.float64)))
    thr = F32(iou_threshold)
    keep = []
    for idx in order:
        if len(keep) >= max_output_size:
            break
        ok = True
        for k in reversed(keep):
            if _iou_f32(boxes[idx], boxes[k]) > thr:
                ok = False
                break
        if ok:
            keep.append(int(idx))
    return np.asarray(keep, np.int64)


def mold_batch(inputs, batch_indices, batch_size, max_batch_size=32):
    """Reference MoldBatch.call (engine/layers/misc.py:231-286): partition rows by
    image (tf.dynamic_partition with 32 slots, :275), pad each partition with -1 to
    max(1, max rows per image) (:235-236, :276-282), stack, keep [:batch_size]."""
    batch_indices = np.asarray(batch_indices, np.int64)
    if max_batch_size is not None and batch_size > 32:
        raise ValueError("MoldBatch supports at most 32 images (misc.py:275)")
    counts = np.bincount(batch_indices, minlength=batch_size) if len(batch_indices) else \
        np.zeros(batch_size, np.int64)
    n = max(1, int(counts.max()) if len(counts) else 1)
    out = np.full((batch_size, n) + inputs.shape[1:], -1, inputs.dtype)
    for b in range(batch_size):
        rows = inputs[batch_indices == b]
        out[b, :len(rows)] = rows
    return out


def dilation2d(x, kernel, padding="SAME"):
    """tf.nn.dilation2d, strides = rates = 1, SAME (call site: semantic.py:283):
    out[b,y,x,c] = max_{dy,dx} in[b, y + dy - pad_top, x + dx - pad_left, c] + kernel[dy,dx,c],
    pad_top = (kh-1)//2, pad_left = (kw-1)//2 (Appendix A 'same' rule); positions outside the input
    do not take part in the max (TF's kernel skips them)."""
    assert padding == "SAME"
    B, H, W, C = x.shape
    kh, kw, kc = kernel.shape
    assert kc == C
    pt, pl = (kh - 1) // 2, (kw - 1) // 2
    out = np.full(x.shape, -np.inf, x.dtype)
    for dy in range(kh):
        for dx in range(kw):
            oy, ox = dy - pt, dx - pl                      # out[y,x] sees in[y+oy, x+ox]
            y0, y1 = max(0, -oy), min(H, H - oy)
            x0, x1 = max(0, -ox), min(W, W - ox)
            if y0 >= y1 or x0 >= x1:
                continue
            cand = x[:, y0 + oy:y1 + oy, x0 + ox:x1 + ox, :] + kernel[dy, dx].astype(x.dtype)
            out[:, y0:y1, x0:x1, :] = np.maximum(out[:, y0:y1, x0:x1, :], cand)
    return out


def erosion2d(x, kernel, padding="SAME"):
    """tf.nn.erosion2d (call site: semantic.py:279).  TF 1.x defines it by duality (nn_ops.py):
    negative(dilation2d(negative(value), reverse(kernel, [0, 1]), padding))."""
    return -dilation2d(-x, kernel[::-1, ::-1], padding)
