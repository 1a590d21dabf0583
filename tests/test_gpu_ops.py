"""GPU parity tests (run with -m gpu on the MI355X box): every C-ABI kernel against the CPU
oracle on the same seeded inputs.  Tolerances: float outputs 1e-4 abs on O(1) data unless noted
(the bar for the whole path is 1e-3, BASELINE.json); index outputs bit-exact."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import masklab as O
from oracle import tfops as T

RNG = np.random.default_rng(11)


def rnd(*shape, scale=1.0):
    return (RNG.normal(size=shape) * scale).astype(np.float32)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    torch.cuda.synchronize()
    return t.cpu().numpy()


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from masklab_hip import _lib
    _lib.check(_lib.load().ml_device_check(), "ml_device_check")


def _conv(packed, x, **kw):
    from masklab_hip import ops
    return ops.conv2d(dev(x), ops.DeviceConv(packed, "cuda"), **kw)


@pytest.fixture(params=["f32", "f32x3"])
def conv_math(request):
    """The dense-conv tests run twice on the same tolerances: exact fp32 products (ML_MATH_F32) and the split-operand
    products on the f16 matrix pipe (ML_MATH_F32X3: fp32 tensors, 22-bit operands, fp32 accumulation)."""
    from masklab_hip import ops
    ops.set_conv_math(request.param)
    yield request.param
    ops.set_conv_math("f32")


# ------------------------------------------------------------------ conv family
@pytest.mark.parametrize("k,cin,cout,stride,padding,dil,act,hw", [
    (1, 64, 256, 1, "valid", 1, "relu", (40, 24)),        # backbone 1x1, N tile 128
    (3, 128, 128, 1, "same", 1, "relu", (33, 35)),        # tower conv, ragged M
    (3, 128, 75, 1, "same", 1, "sigmoid", (16, 16)),      # cls output, N tile 32 x3
    (3, 128, 60, 1, "same", 1, None, (8, 8)),             # loc output, N tile 64
    (1, 512, 32, 1, "valid", 1, None, (16, 16)),          # skip projection
    (1, 640, 128, 1, "valid", 1, None, (4, 4)),           # concat projection
    (3, 2048, 128, 2, "same", 1, "relu", (4, 4)),         # P6 conv (stride 2, same)
    (3, 128, 128, 2, ((0, 1), (0, 1)), 1, "relu", (7, 9)),  # mobilenet P7 style explicit pad
    (1, 128, 5, 1, "same", 1, "sigmoid", (28, 28)),       # mask output
    (1, 256, 512, 2, "valid", 1, None, (16, 16)),         # strided shortcut
    (3, 160, 128, 1, "same", 1, "relu", (16, 16)),        # decoder conv (cin 160)
    (1, 8, 128, 1, "valid", 1, "sigmoid", (1, 1)),        # SE dense (cin 8 < 32)
    (3, 32, 96, 1, "same", 3, "relu6", (12, 12)),         # dilation 3
])
def test_conv2d_dense(k, cin, cout, stride, padding, dil, act, hw, conv_math):
    from masklab_hip import _lib, packing
    B = 2
    x = rnd(B, hw[0], hw[1], cin)
    w, b = rnd(k, k, cin, cout, scale=1.0 / np.sqrt(k * k * cin)), rnd(cout)
    ref = T.conv2d(x.astype(np.float64), w, b, stride, padding, dil)
    ref = {"relu": T.relu, "relu6": T.relu6, "sigmoid": T.sigmoid, None: lambda v: v}[act](ref)
    got = host(_conv(packing.pack_dense(w, b), x, stride=stride, padding=padding, dilation=dil,
                     act=_lib.ACT_BY_NAME[act]))
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-5)


@pytest.mark.parametrize("cin,cout,hw,B,res,act,in_slice", [
    (64, 128, (40, 24), 2, False, "relu", False),       # K = 64: two chunks, FIRST then LAST
    (64, 256, (37, 29), 3, False, "relu", False),       # ragged M (3219 rows: partial last panel), two N tiles
    (128, 256, (32, 32), 2, True, "relu", False),       # bottleneck exit: residual + ReLU
    (256, 128, (33, 31), 2, False, None, True),         # reads a channel slice of a wider buffer, linear
    (256, 512, (16, 16), 2, True, "relu6", False),      # few panels: N tiles split over work units
    (512, 1024, (8, 8), 1, True, "relu", False),        # one panel, 8 N tiles in 8 units
    (96, 128, (20, 20), 1, False, "relu", False),       # K = 96: three chunks
    (128, 384, (40, 40), 2, True, "relu", False),       # 3 N tiles: an odd count cannot be split, one unit per panel
    (64, 768, (64, 64), 1, False, "relu", False),       # 6 N tiles: split in two groups of 3
    (128, 640, (24, 24), 3, True, None, False),         # 5 N tiles, 14 panels (M tail of 64 rows)
    (64, 256, (80, 80), 16, True, "relu", False),       # 800 panels: the split that fills the last round (4 rounds of 2 groups)
])
def test_conv1x1_pipelined_kernel(cin, cout, hw, B, res, act, in_slice):
    """csrc/conv1x1_pipe.hip (tile = 4): the persistent, tile-pipelined 1x1 conv of the ResNeXt bottleneck blocks
    (reference engine/backbone/ResNext.py:199-231) against the oracle, and against the generic kernel (tile = 1)."""
    from masklab_hip import _lib, ops, packing
    x = rnd(B, hw[0], hw[1], cin + (32 if in_slice else 0))
    xin = x[..., 32:] if in_slice else x
    w, b = rnd(1, 1, cin, cout, scale=1.0 / np.sqrt(cin)), rnd(cout)
    r = rnd(B, hw[0], hw[1], cout) if res else None
    ref = T.conv2d(xin.astype(np.float64), w, b, 1, "valid", 1)
    if res:
        ref = ref + r
    ref = {"relu": T.relu, "relu6": T.relu6, None: lambda v: v}[act](ref)
    outs = {}
    for tile in (4, 1):
        dc = ops.DeviceConv(packing.pack_dense(w, b, tile=tile), "cuda")
        got = ops.conv2d(dev(x), dc, padding="valid", act=_lib.ACT_BY_NAME[act], residual=dev(r) if res else None,
                         in_coff=32 if in_slice else 0)
        outs[tile] = host(got)
        np.testing.assert_allclose(outs[tile], ref, rtol=0, atol=3e-5, err_msg=f"tile={tile}")
    np.testing.assert_allclose(outs[4], outs[1], rtol=0, atol=1e-5)          # same products; bias / residual enter the chain at a different place
    # writing into a channel slice of a wider buffer leaves the other channels alone
    buf = torch.full((B, hw[0], hw[1], cout + 128), 7.0, device="cuda")
    dc = ops.DeviceConv(packing.pack_dense(w, b, tile=4), "cuda")
    if not res:
        ops.conv2d(dev(x), dc, padding="valid", act=_lib.ACT_BY_NAME[act], out=buf, out_coff=128,
                   in_coff=32 if in_slice else 0)
        hb = host(buf)
        assert np.all(hb[..., :128] == 7.0)
        np.testing.assert_array_equal(hb[..., 128:], outs[4])


def test_conv1x1_pipelined_kernel_is_the_default_for_short_k():
    from masklab_hip import _lib, ops, packing
    lib = _lib.load()
    x = dev(rnd(1, 64, 64, 128))
    for cout, tile, want in ((256, 0, 1), (256, 1, 0), (75, 0, 0), (256, 4, 1)):
        d, _, _ = ops._conv_desc(x, ops.DeviceConv(packing.pack_dense(rnd(1, 1, 128, cout), None, tile=tile), "cuda"),
                                 padding="valid")
        assert lib.ml_conv2d_uses_pipe(d) == want, (cout, tile)
    # the choice looks at one image's pixel count, never at the batch (an image's result must not depend on its shard)
    for B, hw, want in ((1, 32, 0), (64, 32, 0), (1, 64, 1)):
        d, _, _ = ops._conv_desc(dev(rnd(B, hw, hw, 128)), ops.DeviceConv(packing.pack_dense(rnd(1, 1, 128, 256), None), "cuda"),
                                 padding="valid")
        assert lib.ml_conv2d_uses_pipe(d) == want, (B, hw)
    with pytest.raises(RuntimeError, match="tile = 4"):       # forcing it onto a 3x3 conv is an error, not a fallback
        ops.conv2d(dev(rnd(1, 8, 8, 64)), ops.DeviceConv(packing.pack_dense(rnd(3, 3, 64, 128), None, tile=4), "cuda"))


@pytest.mark.parametrize("seed", range(24))
def test_conv2d_random_shapes(seed, conv_math):
    """Seeded random geometry against the oracle: kernel 1 / 3 / 5 (also non-square maps), stride 1 / 2, dilation 1-3,
    'same' / 'valid', channel counts off the tile grid, optional residual, every activation -- both conv maths."""
    from masklab_hip import _lib, ops, packing
    rng = np.random.default_rng(1000 + seed)
    k = int(rng.choice([1, 3, 3, 5]))
    stride = int(rng.choice([1, 1, 2]))
    dil = 1 if (k == 1 or stride == 2) else int(rng.choice([1, 2, 3]))
    cin = int(rng.choice([8, 24, 32, 40, 64, 96, 136, 256]))
    cout = int(rng.choice([5, 32, 60, 75, 128, 130, 192, 256]))
    B, H, W = int(rng.integers(1, 4)), int(rng.integers(5, 40)), int(rng.integers(5, 40))
    padding = str(rng.choice(["same", "valid"]))
    if padding == "valid" and min(H, W) <= dil * (k - 1):
        padding = "same"
    act = [None, "relu", "relu6", "sigmoid"][int(rng.integers(0, 4))]
    x = (rng.normal(size=(B, H, W, cin))).astype(np.float32)
    w = (rng.normal(size=(k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
    b = rng.normal(size=(cout,)).astype(np.float32) if rng.integers(0, 2) else None
    ref = T.conv2d(x.astype(np.float64), w, b, stride, padding, dil)
    res = None
    if act != "sigmoid" and cout % 4 == 0 and rng.integers(0, 2):
        res = rng.normal(size=ref.shape).astype(np.float32)
        ref = ref + res
    ref = {"relu": T.relu, "relu6": T.relu6, "sigmoid": T.sigmoid, None: lambda v: v}[act](ref)
    dc = ops.DeviceConv(packing.pack_dense(w, b), "cuda")
    got = host(ops.conv2d(dev(x), dc, stride=stride, padding=padding, dilation=dil, act=_lib.ACT_BY_NAME[act],
                          residual=None if res is None else dev(res)))
    assert got.shape == ref.shape, (k, stride, dil, cin, cout, (B, H, W), padding, act)
    np.testing.assert_allclose(got, ref, rtol=0, atol=3e-5, err_msg=str((k, stride, dil, cin, cout, (B, H, W), padding, act)))


def test_conv2d_residual_and_concat_slice(conv_math):
    from masklab_hip import _lib, ops, packing
    x, res = rnd(2, 12, 12, 64), rnd(2, 12, 12, 96)
    w, b = rnd(1, 1, 64, 96, scale=0.1), rnd(96)
    ref = T.relu(T.conv2d(x.astype(np.float64), w, b, padding="valid") + res)
    dc = ops.DeviceConv(packing.pack_dense(w, b), "cuda")
    got = host(ops.conv2d(dev(x), dc, padding="valid", act=_lib.ACT_RELU, residual=dev(res)))
    np.testing.assert_allclose(got, ref, atol=2e-5)
    buf = torch.full((2, 12, 12, 160), 7.0, device="cuda")
    ops.conv2d(dev(x), dc, padding="valid", out=buf, out_coff=32)
    o = host(buf)
    np.testing.assert_allclose(o[..., 32:128], T.conv2d(x.astype(np.float64), w, b, padding="valid"), atol=2e-5)
    assert np.all(o[..., :32] == 7.0) and np.all(o[..., 128:] == 7.0)


def test_conv2d_out_view_concatenated_prediction(conv_math):
    from masklab_hip import ops, packing
    B, nc, pri = 3, 5, 15
    levels = [(8, 8), (4, 4)]
    total = sum(h * w * pri for h, w in levels)
    pred = torch.zeros((B, total, nc), device="cuda")
    off, refs = 0, []
    for (h, w_) in levels:
        x = rnd(B, h, w_, 32)
        w, b = rnd(3, 3, 32, pri * nc, scale=0.05), rnd(pri * nc)
        refs.append(T.conv2d(x.astype(np.float64), w, b).reshape(B, -1, nc))
        ops.conv2d(dev(x), ops.DeviceConv(packing.pack_dense(w, b), "cuda"),
                   out_view=(pred, off * nc, pri * nc, total * nc))
        off += h * w_ * pri
    np.testing.assert_allclose(host(pred), np.concatenate(refs, 1), atol=2e-5)


@pytest.mark.parametrize("k,stride,padding,cout,act", [(7, 2, ((3, 3), (3, 3)), 64, "relu"),
                                                       (3, 2, ((0, 1), (0, 1)), 32, "relu6")])
def test_conv2d_rowspan_stem(k, stride, padding, cout, act, conv_math):
    from masklab_hip import _lib, packing
    img = rnd(2, 64, 48, 3)
    x4 = np.zeros((2, 64, 48, 4), np.float32)
    x4[..., :3] = img
    w, b = rnd(k, k, 3, cout, scale=0.2), rnd(cout)
    ref = {"relu": T.relu, "relu6": T.relu6}[act](T.conv2d(img.astype(np.float64), w, b, stride, padding))
    got = host(_conv(packing.pack_rowspan(w, b), x4, stride=stride, padding=padding, act=_lib.ACT_BY_NAME[act]))
    np.testing.assert_allclose(got, ref, atol=2e-5)


@pytest.mark.parametrize("B,H,W", [(2, 64, 96), (1, 37, 41), (3, 130, 70), (1, 256, 512)])
def test_fused_fp32_stem_and_pool_equals_the_two_kernels(B, H, W, conv_math):
    """csrc/stem_f32.hip, stem_x3.hip (round 4): ZeroPadding2D(3) + 7x7 stride-2 conv + folded BN + ReLU + ZeroPadding2D(1) +
    3x3 stride-2 max-pool in ONE kernel on fp32 tensors (reference engine/backbone/ResNext.py:343-352).  "f32": exact fp32
    products, only those with a non-zero weight, in the generic kernel's pairs and order, from the bias; "f32x3": the generic
    kernel's split-operand steps.  Either way BIT-identical to conv2d + maxpool3x3s2 in that math, and the oracle's values
    within the conv tolerance.  Odd sizes exercise the zero padding on every side and partial pooled tiles; a region of
    mostly negative conv outputs checks the ReLU / zero-padding interplay of the pool."""
    from masklab_hip import _lib, ops, packing
    x = rnd(B, H, W, 3)
    x[0, :, : W // 3] -= 3.0                                   # a region where most conv outputs are cut by the ReLU
    x4 = np.concatenate([x, np.zeros_like(x[..., :1])], -1)
    w, b = rnd(7, 7, 3, 64, scale=0.08), rnd(64)
    dc = ops.DeviceConv(packing.pack_rowspan(w, b), "cuda")
    two = ops.maxpool3x3s2(ops.conv2d(dev(x4), dc, stride=2, padding=((3, 3), (3, 3)), act=_lib.ACT_RELU), pad=1)
    one = ops.stem_pool(dev(x4), dc)
    assert one.dtype == torch.float32 and one.shape == two.shape
    np.testing.assert_array_equal(host(one), host(two))
    conv = T.relu(T.conv2d(x.astype(np.float64), w, b, 2, ((3, 3), (3, 3))))
    ref = T.max_pool(np.pad(conv, ((0, 0), (1, 1), (1, 1), (0, 0))), 3, 2)
    np.testing.assert_allclose(host(one), ref, rtol=0, atol=2e-5)
    with pytest.raises(ValueError):
        ops.stem_pool(dev(x4), ops.DeviceConv(packing.pack_dense(rnd(3, 3, 4, 64), b), "cuda"))


@pytest.mark.parametrize("c,stride,filters", [(4, 1, 128), (8, 2, 256), (16, 1, 512), (32, 2, 1024)])
def test_conv2d_grouped_resnext(c, stride, filters, conv_math):
    from masklab_hip import _lib, packing
    groups = filters // c
    x = rnd(2, 10, 12, filters)
    k = rnd(3, 3, filters, c, scale=1.0 / np.sqrt(9 * c))
    ref = T.relu(O.grouped_conv_fast(x.astype(np.float64), k, groups, c, stride))
    got = host(_conv(packing.pack_grouped(k, groups), x, stride=stride, padding=((1, 1), (1, 1)), act=_lib.ACT_RELU))
    np.testing.assert_allclose(got, ref, atol=2e-5)


@pytest.mark.parametrize("c,stride,filters,hw", [(4, 1, 128, (16, 16)), (4, 1, 128, (13, 21)), (8, 1, 256, (10, 12)),
                                                  (8, 2, 256, (16, 24)), (16, 1, 512, (9, 9)), (16, 2, 512, (8, 8)),
                                                  (4, 2, 128, (31, 17)), (8, 1, 64, (8, 8))])
def test_gconv3x3_mfma4(c, stride, filters, hw):
    """dedicated grouped 3x3 (v_mfma_f32_4x4x1 16-block form) vs the reference's literal spelling"""
    from masklab_hip import _lib, ops, packing
    groups = filters // c
    x = rnd(2, hw[0], hw[1], filters)
    k = rnd(3, 3, filters, c, scale=1.0 / np.sqrt(9 * c))
    b = rnd(filters)
    ref = T.relu(O.grouped_conv_fast(x.astype(np.float64), k, groups, c, stride) + b)
    got = host(ops.gconv3x3(dev(x), dev(packing.pack_grouped_mfma4(k, groups)), dev(b), c, stride=stride,
                            act=_lib.ACT_RELU))
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, atol=2e-5)


@pytest.mark.parametrize("dtype,c,stride,filters,shape", [
    ("f32", 4, 1, 128, (8, 256, 256)),      # 8 tiles per block (32 tile rows of a column in 4 runs)
    ("f32", 8, 2, 256, (8, 256, 256)),      # stride 2: 4 x 8 output tiles, runs of 8
    ("f16", 16, 1, 512, (16, 80, 80)),      # half, 16 x 8 tiles: 5 tile rows in runs of 3 + 2
    ("f16", 8, 1, 256, (16, 152, 150)),     # half, ragged map: partial tiles at the right and bottom edge, runs of 5
])
def test_gconv3x3_blocks_that_walk_several_tiles(dtype, c, stride, filters, shape):
    """Round 4: in a launch that fills the chip a block walks a RUN of tiles down one tile column (weights resident, the
    next tile's halo prefetched under the current tile's MFMAs; csrc/gconv_mfma4.hip column_run) instead of one tile.  The
    values cannot depend on that: an image inside the batch (long runs) equals the same image run alone (one tile per
    block at these sizes) bit for bit, and the oracle within the op tolerance (reference engine/backbone/ResNext.py:212-219)."""
    from masklab_hip import _lib, ops, packing
    B, H, W = shape
    groups = filters // c
    half = dtype == "f16"
    x = rnd(B, H, W, filters)
    if half:
        x = x.astype(np.float16)
    k = rnd(3, 3, filters, c, scale=1.0 / np.sqrt(9 * c))
    b = rnd(filters)
    wg, bd = dev(packing.pack_grouped_mfma4(k, groups)), dev(b)
    xd = dev(x)
    full = ops.gconv3x3(xd, wg, bd, c, stride=stride, act=_lib.ACT_RELU)
    for i in (0, B - 1):
        alone = ops.gconv3x3(xd[i:i + 1].contiguous(), wg, bd, c, stride=stride, act=_lib.ACT_RELU)
        assert torch.equal(full[i:i + 1], alone), i
    kk = k.astype(np.float16).astype(np.float64) if half else k          # the half kernels round the weights to half
    ref = T.relu(O.grouped_conv_fast(x[B - 1:].astype(np.float64), kk, groups, c, stride) + b)
    got = host(full[B - 1:]).astype(np.float64)
    assert got.shape == ref.shape
    if half:
        np.testing.assert_allclose(got, ref, rtol=2e-3, atol=2e-3)
    else:
        np.testing.assert_allclose(got, ref, atol=2e-5)


def test_conv2d_transpose2x2(conv_math):
    from masklab_hip import _lib, packing
    x = rnd(5, 14, 14, 128)
    w, b = rnd(2, 2, 128, 128, scale=0.05), rnd(128)
    ref = T.relu(T.conv2d_transpose_2x2_s2(x.astype(np.float64), w, b))
    got = host(_conv(packing.pack_transpose2x2(w, b), x, act=_lib.ACT_RELU))
    assert got.shape == (5, 28, 28, 128)
    np.testing.assert_allclose(got, ref, atol=2e-5)


@pytest.mark.parametrize("cmid,K,ncls,levels,hw", [
    (128, 128, 3, [(2, 3), (2, 1), (2, 5)], (14, 14)),           # the benchmark's mask head
    (256, 256, 3, [(1, 2), (1, 3)], (14, 14)),                   # Keras default width
    (128, 96, 20, [(3, 4)], (14, 14)),                           # 20 classes: 32-wide table
    (128, 128, 1, [(2, 7), (2, 2), (2, 1), (2, 4)], (14, 14)),
    (128, 64, 5, [(2, 9), (2, 30)], (7, 7)),                     # crop_size 7: RoI maps smaller than a tile row group
    (128, 128, 2, [(1, 3), (1, 2)], (6, 10)),                    # non-square crop
    (128, 128, 3, [(8, 100)], (14, 14))])                        # 800 RoIs: several units per persistent block
def test_deconv2x2_out1x1_fused_tail(cmid, K, ncls, levels, hw):
    """csrc/deconv_out.hip (MaskSubNet tail, instance.py:196-201,226-233): Conv2DTranspose 2x2 s2 + ReLU -> Conv2D 1x1 +
    sigmoid per RoI level with its own weights, all levels in one launch, written into the image-major
    [B, total, 2h, 2w, ncls] tensor at each level's RoI offset; partly filled 128-pixel tiles at every level end."""
    from masklab_hip import _lib, ops, packing
    h, w_ = hw
    B = levels[0][0]
    total = sum(n for _, n in levels)
    out = torch.full((B, total, 2 * h, 2 * w_, ncls), -7.0, device="cuda")
    per_roi = 4 * h * w_ * ncls
    want = np.zeros((B, total, 2 * h, 2 * w_, ncls))
    problems, off = [], 0
    for _, n in levels:
        x = rnd(B * n, h, w_, K)
        wd, bd = rnd(2, 2, cmid, K, scale=0.05), rnd(cmid)
        wo, bo = rnd(1, 1, cmid, ncls, scale=0.1), rnd(ncls)
        t = T.relu(T.conv2d_transpose_2x2_s2(x.astype(np.float64), wd, bd))
        y = T.sigmoid(T.conv2d(t, wo, bo))
        want[:, off:off + n] = y.reshape(B, n, 2 * h, 2 * w_, ncls)
        table, bo_p, cp = packing.pack_out1x1_table(wo, bo)
        problems.append(dict(x=dev(x), dc=ops.DeviceConv(packing.pack_transpose2x2(wd, bd), "cuda"),
                             wo_table=dev(table), bo=dev(bo_p), out=out, out_base=off * per_roi, rois_per_image=n))
        off += n
    ops.deconv2x2_out1x1_multi(problems, ncls, _lib.ACT_RELU, _lib.ACT_SIGMOID)
    np.testing.assert_allclose(host(out), want, atol=2e-5)


def test_deconv2x2_out1x1_rejects_bad_shapes():
    from masklab_hip import _lib, ops, packing
    x = dev(rnd(2, 14, 14, 64))
    wd, wo = rnd(2, 2, 64, 64, scale=0.05), rnd(1, 1, 64, 3)
    table, bo, _ = packing.pack_out1x1_table(wo, None)
    out = torch.zeros((2, 1, 28, 28, 3), device="cuda")
    with pytest.raises(RuntimeError, match="128 or 256"):
        ops.deconv2x2_out1x1_multi([dict(x=x, dc=ops.DeviceConv(packing.pack_transpose2x2(wd, None), "cuda"),
                                         wo_table=dev(table), bo=dev(bo), out=out, out_base=0, rois_per_image=1)],
                                   3, _lib.ACT_RELU, _lib.ACT_SIGMOID)


@pytest.mark.parametrize("k,cin,cout,hw,stride", [(3, 2048, 128, (6, 6), 2), (1, 2048, 128, (16, 16), 1),
                                                  (1, 2048, 128, (1, 1), 1), (3, 128, 128, (8, 8), 1),
                                                  (1, 640, 128, (32, 32), 1), (3, 256, 75, (4, 4), 1)])
def test_conv2d_split_k_small_m(k, cin, cout, hw, stride, conv_math):
    """few output tiles + long K => the split-K path (partials in the workspace, fixed-order reduce)."""
    from masklab_hip import _lib, packing
    x = rnd(2, hw[0], hw[1], cin)
    w, b = rnd(k, k, cin, cout, scale=1.0 / np.sqrt(k * k * cin)), rnd(cout)
    res = None
    ref = T.conv2d(x.astype(np.float64), w, b, stride, "same")
    if stride == 1:
        res = rnd(*ref.shape)
        ref = ref + res
    ref = T.relu(ref)
    from masklab_hip import ops
    dc = ops.DeviceConv(packing.pack_dense(w, b), "cuda")
    got1 = host(ops.conv2d(dev(x), dc, stride=stride, padding="same", act=_lib.ACT_RELU,
                           residual=None if res is None else dev(res)))
    np.testing.assert_allclose(got1, ref, atol=3e-5)
    got2 = host(ops.conv2d(dev(x), dc, stride=stride, padding="same", act=_lib.ACT_RELU,
                           residual=None if res is None else dev(res)))
    np.testing.assert_array_equal(got1, got2)        # deterministic reduction order


def test_conv2d_multi_problem_launch(conv_math):
    """the same conv shape at 5 pyramid levels + a strided-view destination, one launch"""
    from masklab_hip import _lib, ops, packing
    B, nc, pri = 2, 5, 15
    levels = [(16, 16), (8, 8), (4, 4), (2, 2), (1, 1)]
    xs = [rnd(B, h, w_, 128) for h, w_ in levels]
    ws = [(rnd(3, 3, 128, 128, scale=0.03), rnd(128)) for _ in levels]
    outs = ops.conv2d_multi([dict(x=dev(x), dc=ops.DeviceConv(packing.pack_dense(w, b), "cuda"), act=_lib.ACT_RELU)
                             for x, (w, b) in zip(xs, ws)])
    for x, (w, b), o in zip(xs, ws, outs):
        np.testing.assert_allclose(host(o), T.relu(T.conv2d(x.astype(np.float64), w, b)), atol=2e-5)
    total = sum(h * w_ * pri for h, w_ in levels)
    pred = torch.zeros((B, total, nc), device="cuda")
    wo = [(rnd(3, 3, 128, pri * nc, scale=0.03), rnd(pri * nc)) for _ in levels]
    probs, off, refs = [], 0, []
    for x, (w, b), (h, w_) in zip(xs, wo, levels):
        probs.append(dict(x=dev(x), dc=ops.DeviceConv(packing.pack_dense(w, b), "cuda"), act=_lib.ACT_SIGMOID,
                          out_view=(pred, off * nc, pri * nc, total * nc)))
        refs.append(T.sigmoid(T.conv2d(x.astype(np.float64), w, b)).reshape(B, -1, nc))
        off += h * w_ * pri
    ops.conv2d_multi(probs)
    np.testing.assert_allclose(host(pred), np.concatenate(refs, 1), atol=2e-5)


def test_conv2d_rejects_bad_arguments():
    from masklab_hip import ops, packing
    dc = ops.DeviceConv(packing.pack_dense(rnd(1, 1, 64, 8)), "cuda")
    with pytest.raises(ValueError):
        ops.conv2d(dev(rnd(1, 4, 4, 32)), dc)                 # too few input channels
    with pytest.raises(RuntimeError):
        ops.conv2d(torch.zeros(1, 4, 4, 64), dc)              # host tensor: no CPU fallback


# ------------------------------------------------------------------ depthwise / pool / preprocess
@pytest.mark.parametrize("stride,padding,dil,C,hw", [(1, "same", 1, 64, (20, 20)), (2, ((0, 1), (0, 1)), 1, 128, (16, 18)),
                                                     (1, "same", 6, 256, (32, 32)), (1, "same", 18, 64, (32, 32)),
                                                     (1, "same", 12, 32, (8, 8))])
def test_dwconv3x3(stride, padding, dil, C, hw):
    from masklab_hip import _lib, ops, packing
    x = rnd(2, hw[0], hw[1], C)
    k, b = rnd(3, 3, C, 1, scale=0.3), rnd(C)
    ref = T.relu6(T.depthwise_conv2d(x.astype(np.float64), k, stride, padding, dil) + b)
    got = host(ops.dwconv3x3(dev(x), dev(packing.pack_depthwise(k)), dev(b), stride=stride, padding=padding,
                             dilation=dil, act=_lib.ACT_RELU6))
    np.testing.assert_allclose(got, ref, atol=1e-5)


def test_maxpool3x3s2():
    from masklab_hip import ops
    x = np.abs(rnd(2, 17, 20, 64))
    ref = T.max_pool(np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0))), 3, 2)
    np.testing.assert_array_equal(host(ops.maxpool3x3s2(dev(x))), ref)
    xn = rnd(1, 8, 8, 8)          # negative values: the explicit zero pad must win at the border
    refn = T.max_pool(np.pad(xn, ((0, 0), (1, 1), (1, 1), (0, 0))), 3, 2)
    np.testing.assert_array_equal(host(ops.maxpool3x3s2(dev(xn))), refn)


@pytest.mark.parametrize("bt", ["resnext50", "mobilenet"])
@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_preprocess(bt, dtype):
    from masklab_hip.backbone import BackBonePreProcess
    img = RNG.integers(0, 256, (2, 9, 7, 3)).astype(dtype)
    kw = O.PREPROCESS[bt]
    ref = O.backbone_preprocess(img.astype(np.float32), **kw)
    got = host(BackBonePreProcess(**kw)(dev(img)))
    assert got.shape == (2, 9, 7, 4) and np.all(got[..., 3] == 0)
    np.testing.assert_allclose(got[..., :3], ref, atol=1e-6)


def test_preprocess_standardisation_mode():
    """normalize=3 (reference base.py:71-73; note the reference's RGB std order [0.225,0.224,0.229])"""
    from masklab_hip.backbone import BackBonePreProcess
    img = RNG.integers(0, 256, (1, 5, 6, 3)).astype(np.float32)
    for rgb in (True, False):
        ref = O.backbone_preprocess(img, rgb=rgb, mean_shift=True, normalize=3)
        got = host(BackBonePreProcess(rgb=rgb, mean_shift=True, normalize=3)(dev(img)))[..., :3]
        np.testing.assert_allclose(got, ref, rtol=2e-6, atol=1e-6)


# ------------------------------------------------------------------ GroupNormalization (chunk-norm)
@pytest.mark.parametrize("shape,G,relu", [((2, 16, 16, 128), 16, False), ((3, 14, 14, 128), 16, False),
                                          ((2, 32, 32, 32), 16, True), ((2, 2, 2, 128), 32, False),
                                          ((1, 128, 128, 128), 16, True), ((4, 1, 1, 128), 32, False),
                                          ((2, 6, 5, 12), 3, False)])
def test_groupnorm_chunk(shape, G, relu):
    from masklab_hip import ops
    x = rnd(*shape) * 3 + 1.5
    gamma, beta = RNG.uniform(0.5, 1.5, shape[-1]).astype(np.float32), rnd(shape[-1])
    ref = T.group_norm(x.astype(np.float64), gamma, beta, G)
    np.testing.assert_allclose(ref, T.group_norm_flat(x, gamma, beta, G), atol=1e-9)
    if relu:
        ref = T.relu(ref)
    got = host(ops.groupnorm_chunk(dev(x), dev(gamma), dev(beta), G, relu=relu))
    np.testing.assert_allclose(got, ref, atol=2e-5)
    xin = dev(x)
    ops.groupnorm_chunk(xin, dev(gamma), dev(beta), G, relu=relu, out=xin)      # in place
    np.testing.assert_allclose(host(xin), ref, atol=2e-5)


def test_groupnorm_into_concat_slice_and_large_mean():
    from masklab_hip import ops
    x = rnd(2, 8, 8, 32) + 300.0            # mean >> std: the fp64 statistics must not cancel
    gamma, beta = RNG.uniform(0.5, 1.5, 32).astype(np.float32), rnd(32)
    ref = T.relu(T.group_norm(x.astype(np.float64), gamma, beta, 16))
    buf = torch.full((2, 8, 8, 160), -5.0, device="cuda")
    ops.groupnorm_chunk(dev(x), dev(gamma), dev(beta), 16, relu=True, out=buf, out_coff=128)
    o = host(buf)
    np.testing.assert_allclose(o[..., 128:], ref, atol=2e-3)
    assert np.all(o[..., :128] == -5.0)


def test_groupnorm_error_messages():
    from masklab_hip import GroupNormalization
    with pytest.raises(ValueError, match="cannot be more than the number of channels"):
        GroupNormalization(groups=32).build((None, 4, 4, 16))
    with pytest.raises(ValueError, match="must be a multiple of the number of channels"):
        GroupNormalization(groups=5).build((None, 4, 4, 16))


# ------------------------------------------------------------------ resampling / reductions
@pytest.mark.parametrize("hw,ohw", [((4, 4), (8, 8)), ((8, 8), (15, 17)), ((1, 1), (16, 16)), ((32, 32), (128, 128)),
                                    ((5, 7), (5, 7))])
def test_resize_bilinear_align_corners(hw, ohw):
    from masklab_hip import ops
    x, add = rnd(2, hw[0], hw[1], 128), rnd(2, ohw[0], ohw[1], 128)
    ref = T.resize_bilinear_align_corners(x.astype(np.float64), *ohw)
    np.testing.assert_allclose(host(ops.resize_bilinear_ac(dev(x), *ohw)), ref, atol=1e-5)
    lat = dev(add)
    ops.resize_bilinear_ac(dev(x), *ohw, add=lat, out=lat)           # FPN: in-place add
    np.testing.assert_allclose(host(lat), ref + add, atol=1e-5)


def test_global_mean_and_scale():
    from masklab_hip import ops
    x = rnd(3, 9, 11, 2048) + 2
    np.testing.assert_allclose(host(ops.global_mean(dev(x)))[:, 0, 0], x.astype(np.float64).mean((1, 2)), atol=1e-5)
    s = rnd(3, 1, 1, 2048)
    np.testing.assert_allclose(host(ops.scale_channels_(dev(x), dev(s))), x * s, atol=1e-6)


def test_groupnorm_multi_equals_single_launches():
    """ml_groupnorm_multi_f32: the five pyramid levels of a tower depth (two-pass P3..P5, register-resident P6 / P7) and
    a non-pixel-aligned RoI map in ONE launch pair -- bit-identical to the single-problem launches, and to the oracle."""
    from masklab_hip import ops
    shapes = [(2, 64, 64, 128, 16), (2, 32, 32, 128, 16), (2, 16, 16, 128, 16), (2, 8, 8, 128, 16), (2, 4, 4, 128, 16),
              (9, 14, 14, 128, 16), (2, 16, 16, 128, 32),
              (1, 128, 128, 128, 16),      # chunk 131 072 floats
              (2, 80, 80, 128, 16)]        # chunk 51 200: slices that do not divide it
    xs = [rnd(*s[:4]) + 0.5 for s in shapes]
    gb = [(RNG.uniform(0.5, 1.5, s[3]).astype(np.float32), rnd(s[3])) for s in shapes]
    single = [host(ops.groupnorm_chunk(dev(x), dev(g), dev(b), s[4])) for x, (g, b), s in zip(xs, gb, shapes)]
    multi = ops.groupnorm_chunk_multi([dict(x=dev(x), gamma=dev(g), beta=dev(b), groups=s[4])
                                       for x, (g, b), s in zip(xs, gb, shapes)])
    for x, (g, b), s, one, m in zip(xs, gb, shapes, single, multi):
        np.testing.assert_array_equal(host(m), one)
        np.testing.assert_allclose(one, T.group_norm(x.astype(np.float64), g, b, s[4]), atol=2e-5)
    # data whose mean dwarfs its spread (fp64 running sums; a float4 is folded in fp32 before it joins them, so the
    # variance carries ~mean^2 * 2^-24 of rounding noise per quad: 300 is the regime the 2e-3 bar covers)
    big = (rnd(1, 128, 128, 32) + 300.0).astype(np.float32)
    g32, b32 = RNG.uniform(0.5, 1.5, 32).astype(np.float32), rnd(32)
    got = host(ops.groupnorm_chunk(dev(big), dev(g32), dev(b32), 16))
    np.testing.assert_allclose(got, T.group_norm(big.astype(np.float64), g32, b32, 16), atol=2e-3)
    # in place and with a fused ReLU into a concat slice
    x0 = dev(xs[0].copy())
    buf = torch.full((2, 32, 32, 160), 7.0, device="cuda")
    outs = ops.groupnorm_chunk_multi([dict(x=x0, gamma=dev(gb[0][0]), beta=dev(gb[0][1]), groups=16, out=x0),
                                      dict(x=dev(xs[1]), gamma=dev(gb[1][0]), beta=dev(gb[1][1]), groups=16, relu=True,
                                           out=buf, out_coff=32)])
    np.testing.assert_array_equal(host(outs[0]), single[0])
    hb = host(buf)
    assert np.all(hb[..., :32] == 7.0)
    np.testing.assert_array_equal(hb[..., 32:], np.maximum(single[1], 0.0))


def test_groupnorm_statistics_from_the_conv_epilogue(conv_math):
    """ml_conv2d_desc.gn_partials / ml_gn_desc.partials: the head convs that feed a GroupNormalization
    (engine/layers/detection.py:120-125, semantic.py:205-213) also write (sum, sum of squares) of every 128-row tile they
    store; the GroupNorm apply pass adds a chunk's tiles in tile order instead of re-reading the tensor.  Same values as
    conv -> two-pass GroupNorm to fp32 rounding of the statistics, and the oracle's; a launch too small for the rule is
    refused loudly; the tower helper picks the form by itself."""
    from masklab_hip import _lib, ops, packing
    from masklab_hip.keras_like import Conv2D
    from masklab_hip.layers.detection import _TowerMixin
    from masklab_hip.normalization import GroupNormalization
    B, H, W = 3, 128, 128                         # 384 tiles; chunk = 1024 pixels = 8 tiles
    x = rnd(B, H, W, 128)
    w, b = rnd(3, 3, 128, 128, scale=0.03), rnd(128)
    gamma, beta = RNG.uniform(0.5, 1.5, 128).astype(np.float32), rnd(128)
    dc = ops.DeviceConv(packing.pack_dense(w, b), "cuda")
    plain = ops.conv2d(dev(x), dc, act=_lib.ACT_RELU)
    want = host(ops.groupnorm_chunk(plain, dev(gamma), dev(beta), 16))
    part = torch.full((B * H * W // 128, 4, 2), float("nan"), dtype=torch.float64, device="cuda")
    y = ops.conv2d(dev(x), dc, act=_lib.ACT_RELU, gn_partials=part)
    np.testing.assert_array_equal(host(y), host(plain))                     # the conv's own output is untouched
    yh = host(y).astype(np.float64).reshape(-1, 128 * 128)
    np.testing.assert_allclose(host(part)[..., 0].sum(1), yh.sum(1), rtol=1e-7)      # (a float4 is folded in fp32 first)
    np.testing.assert_allclose(host(part)[..., 1].sum(1), (yh * yh).sum(1), rtol=1e-6)
    (got,) = ops.groupnorm_chunk_multi([dict(x=y, gamma=dev(gamma), beta=dev(beta), groups=16, out=y, partials=(part, 32))])
    np.testing.assert_allclose(host(got), want, rtol=0, atol=2e-6)
    ref = T.group_norm(T.relu(T.conv2d(x.astype(np.float64), w, b)), gamma, beta, 16)
    np.testing.assert_allclose(host(got), ref, atol=3e-5)
    with pytest.raises(RuntimeError, match="gn_partials"):                  # 64 tiles: the library would narrow / split this launch
        ops.conv2d(dev(x[:1, :64]), dc, act=_lib.ACT_RELU, gn_partials=part)
    # the tower helper: fused for the big level, the old two-pass / one-pass forms for the small ones, same results
    conv, gn = Conv2D(128, (3, 3), activation='relu', padding='same', name="t/conv0"), GroupNormalization(16, name="t/gn0")
    conv.build((None, None, None, 128)); gn.build((None, None, None, 128))
    wd = {"t/conv0/kernel": w, "t/conv0/bias": b, "t/gn0/gamma": gamma, "t/gn0/beta": beta}
    conv.load_weights(wd, torch.device("cuda:0")); gn.load_weights(wd, torch.device("cuda:0"))
    xs = [rnd(B, 128, 128, 128), rnd(B, 64, 64, 128), rnd(B, 16, 16, 128)]
    ops.PROFILE = []
    outs = _TowerMixin._run_towers_multi([[conv, gn]] * 3, [dev(v) for v in xs])
    recs, ops.PROFILE = ops.PROFILE, None
    for v, o in zip(xs, outs):
        np.testing.assert_allclose(host(o), T.group_norm(T.relu(T.conv2d(v.astype(np.float64), w, b)), gamma, beta, 16), atol=3e-5)
    single = host(_TowerMixin._run_tower([conv, gn], dev(xs[0])))
    np.testing.assert_array_equal(single, host(outs[0]))                    # one problem or five: the same tile sums
