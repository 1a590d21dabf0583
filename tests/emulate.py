"""NumPy emulation of the ADDRESSING of csrc/conv_mfma.hip (test helper, CPU only).

It walks the packed weight matrix and the NHWC input exactly the way the kernel does (taps,
span / span_pad, cpp_shift pixel spans, grouped input windows per 32-wide N tile, bias /
residual / activation / 2x2 pixel-shuffle epilogue).  Comparing it with the oracle's conv
validates masklab_hip/packing.py and the descriptor logic without a GPU."""
import numpy as np

from masklab_hip.packing import PackedConv, ntile_for, resolve_padding


def _act(v, act):
    if act == "relu":
        return np.maximum(v, 0)
    if act == "relu6":
        return np.minimum(np.maximum(v, 0), 6)
    if act == "sigmoid":
        return 1.0 / (1.0 + np.exp(-v))
    return v


def emulate_conv(p: PackedConv, x, stride=1, padding="same", dilation=1, act=None, residual=None, in_coff=0):
    x = np.ascontiguousarray(x, np.float64)
    B, H, W, cs = x.shape
    Ho, Wo, pt, pl = resolve_padding(H, W, p.kh_real, p.kw_real, stride, dilation, padding)
    flat = x.reshape(-1)
    M = B * Ho * Wo
    m = np.arange(M)
    b = m // (Ho * Wo)
    r = m % (Ho * Wo)
    oy, ox = r // Wo, r % Wo
    bn = ntile_for(p.cout, p.tile)
    wgt = p.wgt.astype(np.float64)
    ktot = p.KH * p.KW * p.span_pad
    assert wgt.shape == (p.n_pad, ktot)
    out = np.zeros((M, p.n_pad))
    c = np.arange(p.span_pad)
    px = c >> p.cpp_shift
    for nt in range(p.n_pad // bn):
        gofs = in_coff + nt * p.group_cin_step
        cols = slice(nt * bn, (nt + 1) * bn)
        for kh in range(p.KH):
            for kw in range(p.KW):
                iy = oy * stride - pt + kh * dilation
                ix = ox * stride - pl + kw * dilation
                ok = ((iy >= 0) & (iy < H))[:, None] & ((ix[:, None] + px[None]) >= 0) & \
                     ((ix[:, None] + px[None]) < W) & (c < p.span)[None]
                off = ((b * H + iy) * W + ix)[:, None] * cs + gofs + c[None]
                a = np.where(ok, flat[np.clip(off, 0, flat.size - 1)], 0.0)
                k0 = (kh * p.KW + kw) * p.span_pad
                out[:, cols] += a @ wgt[cols, k0:k0 + p.span_pad].T
    out = out[:, :p.cout]
    co = p.cout // 4 if p.shuffle2x2 else p.cout
    if p.bias is not None:
        out = out + np.tile(p.bias.astype(np.float64), 4 if p.shuffle2x2 else 1)
    if residual is not None:
        out = out + residual.reshape(M, -1)
    out = _act(out, act)
    if p.shuffle2x2:
        o = out.reshape(B, Ho, Wo, 2, 2, co)           # column = (a*2+b)*co + o
        return o.transpose(0, 1, 3, 2, 4, 5).reshape(B, 2 * Ho, 2 * Wo, co)
    return out.reshape(B, Ho, Wo, co)
