"""CPU tests: weight packing + conv descriptor addressing (emulated) against the oracle."""
import numpy as np
import pytest

from emulate import emulate_conv
from masklab_hip import packing
from oracle import masklab as O
from oracle import tfops as T

RNG = np.random.default_rng(7)


def rnd(*shape):
    return RNG.normal(size=shape).astype(np.float32)


@pytest.mark.parametrize("k,cin,cout,stride,padding,dil", [
    (1, 64, 128, 1, "valid", 1), (3, 32, 75, 1, "same", 1), (3, 128, 60, 2, "same", 1),
    (3, 8, 5, 1, "same", 1), (1, 160, 3, 1, "same", 1), (3, 36, 40, 2, ((0, 1), (0, 1)), 1),
    (1, 64, 96, 2, "valid", 1), (3, 32, 33, 1, "same", 2),
])
def test_dense_pack_matches_oracle(k, cin, cout, stride, padding, dil):
    x = rnd(2, 9, 11, cin)
    w, b = rnd(k, k, cin, cout), rnd(cout)
    ref = T.conv2d(x.astype(np.float64), w, b, stride, padding, dil)
    got = emulate_conv(packing.pack_dense(w, b), x, stride, padding, dil)
    np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-9)


def test_dense_pack_reads_channel_slice():
    x = rnd(1, 6, 6, 64)
    w = rnd(3, 3, 32, 16)
    ref = T.conv2d(x[..., 32:].astype(np.float64), w)
    got = emulate_conv(packing.pack_dense(w), x, in_coff=32)
    np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("k,stride,padding,cout", [(7, 2, ((3, 3), (3, 3)), 64), (3, 2, ((0, 1), (0, 1)), 32),
                                                   (3, 1, "same", 8)])
def test_rowspan_stem_matches_oracle(k, stride, padding, cout):
    img = rnd(2, 18, 22, 3)
    x4 = np.zeros((2, 18, 22, 4), np.float32)
    x4[..., :3] = img
    w, b = rnd(k, k, 3, cout), rnd(cout)
    ref = T.conv2d(img.astype(np.float64), w, b, stride, padding)
    p = packing.pack_rowspan(w, b)
    assert p.cpp_shift == 2 and p.KW == 1 and p.span == 4 * k
    got = emulate_conv(p, x4, stride, padding)
    np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("c,stride", [(4, 1), (8, 2), (16, 1), (32, 2)])
def test_grouped_pack_matches_reference_spelling(c, stride):
    groups = 32 if c <= 8 else 4
    if c * groups % 32:
        groups = 32 // c * 2
    filters = groups * c
    x = rnd(2, 8, 10, filters)
    k = rnd(3, 3, filters, c)
    ref = O.grouped_conv_literal(x.astype(np.float64), k, groups, c, stride)
    fast = O.grouped_conv_fast(x.astype(np.float64), k, groups, c, stride)
    np.testing.assert_allclose(fast, ref, rtol=1e-9, atol=1e-9)
    dense = T.conv2d(x.astype(np.float64), _block_diag(packing.grouped_dw_to_dense(k, groups), groups, c),
                     None, stride, ((1, 1), (1, 1)))
    np.testing.assert_allclose(dense, ref, rtol=1e-9, atol=1e-9)
    got = emulate_conv(packing.pack_grouped(k, groups), x, stride, ((1, 1), (1, 1)))
    np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-9)


def _block_diag(wg, groups, c):
    kh, kw, _, filters = wg.shape
    full = np.zeros((kh, kw, filters, filters), wg.dtype)
    for g in range(groups):
        full[:, :, g * c:(g + 1) * c, g * c:(g + 1) * c] = wg[:, :, :, g * c:(g + 1) * c]
    return full


def test_transpose2x2_pack_matches_oracle():
    x = rnd(3, 5, 7, 32)
    w, b = rnd(2, 2, 24, 32), rnd(24)
    ref = T.relu(T.conv2d_transpose_2x2_s2(x.astype(np.float64), w, b))
    got = emulate_conv(packing.pack_transpose2x2(w, b), x, act="relu")
    np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-9)


def test_bn_fold_matches_conv_then_bn():
    x = rnd(1, 6, 6, 16)
    w = rnd(3, 3, 16, 8)
    g, beta, mean = RNG.uniform(0.5, 1.5, 8), rnd(8), rnd(8)
    var = RNG.uniform(0.5, 1.5, 8)
    ref = T.batch_norm(T.conv2d(x.astype(np.float64), w), g, beta, mean, var, 1e-3)
    k, b = packing.fold_bn(w, None, g, beta, mean, var, 1e-3)
    np.testing.assert_allclose(T.conv2d(x.astype(np.float64), k, b), ref, rtol=1e-5, atol=1e-5)


def test_residual_and_activation_epilogue():
    x, res = rnd(1, 4, 4, 32), rnd(1, 4, 4, 16)
    w, b = rnd(1, 1, 32, 16), rnd(16)
    ref = T.relu(T.conv2d(x.astype(np.float64), w, b) + res)
    got = emulate_conv(packing.pack_dense(w, b), x, padding="valid", act="relu", residual=res)
    np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-9)


def test_ntile_matches_library():
    from masklab_hip import _lib
    lib = _lib.load()
    for cout in (1, 3, 5, 32, 33, 60, 64, 65, 75, 96, 97, 128, 256, 2048):
        for tile in (0, 1, 2, 3):
            assert lib.ml_conv2d_ntile(cout, tile) == packing.ntile_for(cout, tile)


@pytest.mark.parametrize("pack", ["dense", "rowspan", "grouped"])
def test_split_operand_weights_f32x3(pack):
    """DeviceConv.wgt_x3 (ML_MATH_F32X3, include/masklab_hip.h): every 32-float chunk of a packed row becomes 32 halves
    hi(w) then 32 halves 2^11 (w - hi(w)) -- same bytes, same strides, any packing; hi + 2^-11 lo gives every weight back
    to 2^-22 (2^-36 absolute below the smallest normal half), zeros stay zeros, and the emulated conv on the
    reconstructed weights matches the oracle like the fp32 packing does."""
    torch = pytest.importorskip("torch")
    from masklab_hip import ops
    rng = np.random.default_rng(4)
    if pack == "dense":
        w = (rng.normal(size=(3, 3, 40, 72)) * 0.05).astype(np.float32)
        p = packing.pack_dense(w, rng.normal(size=(72,)).astype(np.float32))
    elif pack == "rowspan":
        w = (rng.normal(size=(7, 7, 3, 64)) * 0.2).astype(np.float32)
        p = packing.pack_rowspan(w, None)
    else:
        w = (rng.normal(size=(3, 3, 128, 4)) * 0.1).astype(np.float32)
        p = packing.pack_grouped(w, 32)
    dc = ops.DeviceConv(p, "cpu")
    a, s = dc.wgt.numpy(), dc.wgt_x3.numpy()
    assert a.shape == s.shape and s.dtype == np.float32 and s.flags["C_CONTIGUOUS"]
    h = s.view(np.float16).reshape(a.shape[0], -1, 64).astype(np.float64)
    assert np.isfinite(h).all()
    back = (h[..., :32] + h[..., 32:] * 2.0 ** -11).reshape(a.shape)
    assert np.all(np.abs(back - a) <= np.maximum(np.abs(a) * 2.0 ** -22, 2.0 ** -36))
    assert np.all(back[a == 0] == 0)
    if pack == "dense":
        import dataclasses
        x = rng.normal(size=(1, 9, 11, 40)).astype(np.float32)
        ref = T.conv2d(x.astype(np.float64), w, p.bias)
        got = emulate_conv(dataclasses.replace(p, wgt=back.astype(np.float32)), x)
        np.testing.assert_allclose(got, ref, atol=2e-6)
