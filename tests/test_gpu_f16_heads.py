"""GPU parity of fp16 STORAGE in the HEADS (BASELINE configs[4]: "fp16 MFMA path"; VERDICT r02 row *): FPN, towers,
mask head, ASPP / decoder and their GroupNormalizations read and write IEEE half in `ops.set_conv_math("f16s")`;
accumulation, statistics, bias and activation are fp32 with ONE rounding at the store; cls_pred / loc_pred / roi_boxes /
roi_masks / seg_pred stay fp32.

Kernel-level bar (as tests/test_gpu_f16_storage.py): the oracle op evaluated in fp64 on the SAME half-rounded
operands, rounded once to half: equal to within one half ulp-step (rtol 2^-10, atol 1e-4).
Model-level bar: against the fp32 oracle forward within 3e-2 on every float output AND the same detections
(reference engine/metrics.py:109-165 DetectionIOUMetric F = 1.0) on a fixture whose threshold sits in a score gap much
wider than the fp16 deviation.  -m gpu."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import masklab as O
from oracle import tfops as T

RNG = np.random.default_rng(31)
HALF_RTOL, HALF_ATOL = 2.0 ** -10, 1e-4
F16_MODEL_TOL = 3e-2


def rnd(*shape, scale=1.0):
    return (RNG.normal(size=shape) * scale).astype(np.float32)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    torch.cuda.synchronize()
    return t.cpu().numpy()


def h64(a):
    return a.astype(np.float16).astype(np.float64)


def to_half(a):
    return a.astype(np.float16)


def close_half(got, ref64, atol=HALF_ATOL):
    """`got` (half tensor from the GPU) vs the fp64 reference rounded once to half."""
    np.testing.assert_allclose(got.astype(np.float32), ref64.astype(np.float16).astype(np.float32), rtol=HALF_RTOL, atol=atol)


@pytest.fixture(scope="module", autouse=True)
def _f16s_mode():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from masklab_hip import ops
    ops.set_conv_math("f16s")
    yield
    ops.set_conv_math("f32")


ACT = {"relu": T.relu, "relu6": T.relu6, None: lambda v: v, "sigmoid": T.sigmoid}


# ------------------------------------------------------------------ generic implicit-GEMM conv on half tensors
@pytest.mark.parametrize("k,cin,cout,stride,padding,dil,act,hw,B", [
    (3, 128, 128, 1, "same", 1, "relu", (40, 40), 2),       # tower conv: 128x128 tiles, 18 chunks of 64
    (3, 128, 128, 1, "same", 1, "relu", (13, 9), 3),        # few tiles: narrow tile + split-K, half output from the reduce
    (3, 160, 128, 1, "same", 1, "relu", (24, 20), 2),       # decoder: 160 channels = 2.5 chunks per tap (zero-padded to 192)
    (3, 128, 128, 2, "same", 1, "relu", (17, 21), 2),       # P6 / P7: stride 2, TF 'same' (asymmetric on even sizes)
    (3, 256, 256, 1, "same", 1, None, (16, 16), 1),         # Keras-default width
    (1, 2048, 128, 1, "same", 1, None, (8, 8), 2),          # C5 lateral of a small image: few tiles, K = 2048 -> split-K
    (1, 640, 128, 1, "same", 1, None, (20, 20), 2),         # ASPP concat projection
    (1, 512, 32, 1, "same", 1, None, (24, 24), 1),          # skip projection: cout 32 (narrow N tile)
    (3, 128, 64, 1, "same", 2, "relu6", (12, 12), 1),       # dilation, 64-wide tile
    (3, 72, 128, 1, "valid", 1, "relu", (10, 11), 1),       # span 72: not a multiple of the 64-deep chunk
])
def test_conv2d_generic_half_storage(k, cin, cout, stride, padding, dil, act, hw, B):
    from masklab_hip import _lib, ops, packing
    x = to_half(rnd(B, hw[0], hw[1], cin))
    w, b = rnd(k, k, cin, cout, scale=1.0 / np.sqrt(k * k * cin)), rnd(cout)
    ref = ACT[act](T.conv2d(x.astype(np.float64), h64(w), b.astype(np.float64), stride, padding, dil))
    dc = ops.DeviceConv(packing.pack_dense(w, b), "cuda")
    got = ops.conv2d(dev(x), dc, stride=stride, padding=padding, dilation=dil, act=_lib.ACT_BY_NAME[act])
    assert got.dtype == torch.float16 and tuple(got.shape) == ref.shape
    close_half(host(got), ref)
    got32 = ops.conv2d(dev(x), dc, stride=stride, padding=padding, dilation=dil, act=_lib.ACT_BY_NAME[act],
                       out_dtype=torch.float32)                              # fp32 destination: no output rounding
    assert got32.dtype == torch.float32
    np.testing.assert_allclose(host(got32), ref, rtol=1e-5, atol=2e-5)


def test_conv2d_half_multi_problem_and_prediction_view():
    """The towers on half tensors: five pyramid levels of one depth in ONE launch (half out), then the output convs of
    all levels written straight into the fp32 [B, A, d] prediction through per-image strided views (sigmoid = the
    generic epilogue, linear = the fast one) -- engine/layers/detection.py:109-130,138-140,197-212."""
    from masklab_hip import _lib, ops, packing
    B, sizes = 2, [(16, 16), (8, 8), (4, 4), (2, 2), (1, 1)]
    xs = [to_half(rnd(B, h, w, 128)) for h, w in sizes]
    ws = [(rnd(3, 3, 128, 128, scale=0.03), rnd(128)) for _ in sizes]
    outs = ops.conv2d_multi([dict(x=dev(x), dc=ops.DeviceConv(packing.pack_dense(w, b), "cuda"), act=_lib.ACT_RELU)
                             for x, (w, b) in zip(xs, ws)])
    for o, x, (w, b) in zip(outs, xs, ws):
        assert o.dtype == torch.float16
        close_half(host(o), T.relu(T.conv2d(x.astype(np.float64), h64(w), b.astype(np.float64))))
    for act, d in (("sigmoid", 5), (None, 4)):
        npri = 15
        total = sum(h * w for h, w in sizes) * npri
        pred = torch.full((B, total, d), -3.0, dtype=torch.float32, device="cuda")
        wo = [(rnd(3, 3, 128, npri * d, scale=0.03), rnd(npri * d)) for _ in sizes]
        problems, off = [], 0
        for x, (w, b), (h, w_) in zip(xs, wo, sizes):
            problems.append(dict(x=dev(x), dc=ops.DeviceConv(packing.pack_dense(w, b), "cuda"), act=_lib.ACT_BY_NAME[act],
                                 out_view=(pred, off * d, npri * d, total * d)))
            off += h * w_ * npri
        ops.conv2d_multi(problems)
        want = np.concatenate([ACT[act](T.conv2d(x.astype(np.float64), h64(w), b.astype(np.float64))).reshape(B, -1, d)
                               for x, (w, b) in zip(xs, wo)], axis=1)
        np.testing.assert_allclose(host(pred), want, rtol=1e-5, atol=2e-5)


def test_conv2d_half_rejects_what_it_does_not_implement():
    from masklab_hip import ops, packing
    x = dev(to_half(rnd(1, 8, 8, 128)))
    dc = ops.DeviceConv(packing.pack_dense(rnd(3, 3, 128, 128), None), "cuda")
    with pytest.raises(RuntimeError, match="residual"):
        ops.conv2d(x, dc, residual=dev(to_half(rnd(1, 8, 8, 128))))
    with pytest.raises(ValueError, match="float16 residual"):
        ops.conv2d(x, dc, residual=dev(rnd(1, 8, 8, 128)))


# ------------------------------------------------------------------ GroupNormalization on half tensors
@pytest.mark.parametrize("shape,G,relu", [((2, 40, 40, 128), 16, False),      # two-pass, hot vector form (C/G = 8)
                                          ((3, 14, 14, 128), 16, False),      # one-pass: chunk 1568 halves = 12.25 pixels
                                          ((2, 32, 32, 32), 16, True),        # C/G = 2: per-element gamma index
                                          ((2, 10, 10, 128), 32, False),      # P6_norm: C/G = 4 < the 8-wide vector
                                          ((1, 96, 96, 160), 16, True),       # decoder concat: C/G = 10
                                          ((2, 5, 5, 128), 32, False),        # chunk 100 halves: scalar path
                                          ((2, 6, 5, 12), 3, False)])
def test_groupnorm_half_storage(shape, G, relu):
    from masklab_hip import ops
    x = to_half(rnd(*shape) * 3 + 1.5)
    gamma, beta = RNG.uniform(0.5, 1.5, shape[-1]).astype(np.float32), rnd(shape[-1])
    ref = T.group_norm(x.astype(np.float64), gamma, beta, G)
    if relu:
        ref = T.relu(ref)
    got = ops.groupnorm_chunk(dev(x), dev(gamma), dev(beta), G, relu=relu)
    assert got.dtype == torch.float16
    close_half(host(got), ref, atol=2e-4)
    xin = dev(x)
    ops.groupnorm_chunk(xin, dev(gamma), dev(beta), G, relu=relu, out=xin)      # in place
    close_half(host(xin), ref, atol=2e-4)


def test_groupnorm_half_multi_and_concat_slice():
    from masklab_hip import ops
    shapes = [(2, 64, 64, 128, 16), (2, 32, 32, 128, 16), (2, 8, 8, 128, 16), (2, 4, 4, 128, 16), (9, 14, 14, 128, 16)]
    xs = [to_half(rnd(*s[:4]) + 0.5) for s in shapes]
    gb = [(RNG.uniform(0.5, 1.5, s[3]).astype(np.float32), rnd(s[3])) for s in shapes]
    single = [host(ops.groupnorm_chunk(dev(x), dev(g), dev(b), s[4])) for x, (g, b), s in zip(xs, gb, shapes)]
    multi = ops.groupnorm_chunk_multi([dict(x=dev(x), gamma=dev(g), beta=dev(b), groups=s[4])
                                       for x, (g, b), s in zip(xs, gb, shapes)])
    for x, (g, b), s, one, m in zip(xs, gb, shapes, single, multi):
        assert m.dtype == torch.float16
        np.testing.assert_array_equal(host(m), one)                             # same arithmetic per problem
        close_half(one, T.group_norm(x.astype(np.float64), g, b, s[4]), atol=2e-4)
    buf = torch.full((2, 32, 32, 160), 7.0, dtype=torch.float16, device="cuda")
    ops.groupnorm_chunk(dev(xs[1]), dev(gb[1][0]), dev(gb[1][1]), 16, relu=True, out=buf, out_coff=128 - 96)
    hb = host(buf)
    assert np.all(hb[..., :32] == 7.0)
    np.testing.assert_array_equal(hb[..., 32:], np.maximum(single[1], 0))
    with pytest.raises(ValueError, match="dtype"):
        ops.groupnorm_chunk(dev(xs[1]), dev(gb[1][0]), dev(gb[1][1]), 16, out=torch.zeros((2, 32, 32, 128), device="cuda"))


# ------------------------------------------------------------------ byte movers with arithmetic
@pytest.mark.parametrize("hw,ohw", [((4, 4), (8, 8)), ((8, 8), (15, 17)), ((1, 1), (16, 16)), ((20, 20), (40, 40))])
def test_resize_bilinear_half_storage(hw, ohw):
    from masklab_hip import ops
    x, add = to_half(rnd(2, hw[0], hw[1], 128)), to_half(rnd(2, ohw[0], ohw[1], 128))
    ref = T.resize_bilinear_align_corners(x.astype(np.float64), *ohw)
    close_half(host(ops.resize_bilinear_ac(dev(x), *ohw)), ref)
    lat = dev(add)
    ops.resize_bilinear_ac(dev(x), *ohw, add=lat, out=lat)           # FPN: in-place add
    close_half(host(lat), ref + add.astype(np.float64))
    cat = torch.zeros((2, ohw[0], ohw[1], 160), dtype=torch.float16, device="cuda")
    ops.resize_bilinear_ac(dev(x), *ohw, out=cat, out_coff=32)       # decoder: into the concat buffer
    close_half(host(cat)[..., 32:], ref)
    assert not host(cat)[..., :32].any()


@pytest.mark.parametrize("dil,C,hw", [(1, 128, (12, 12)), (6, 2048, (10, 10)), (12, 512, (20, 16)), (18, 64, (9, 30))])
def test_dwconv3x3_and_global_mean_half_storage(dil, C, hw):
    from masklab_hip import ops, packing
    x = to_half(rnd(2, hw[0], hw[1], C))
    k = rnd(3, 3, C, 1, scale=0.3)
    ref = T.depthwise_conv2d(x.astype(np.float64), k.astype(np.float64), 1, "same", dil)
    got = ops.dwconv3x3(dev(x), dev(packing.pack_depthwise(k)), None, dilation=dil)
    assert got.dtype == torch.float16
    close_half(host(got), ref)
    m = ops.global_mean(dev(x))
    assert m.dtype == torch.float16 and tuple(m.shape) == (2, 1, 1, C)
    close_half(host(m)[:, 0, 0], x.astype(np.float64).mean((1, 2)))


def test_roi_crop_half_storage():
    from masklab_hip.layers import PyramidRoiAlign
    B, H, W = 3, 256, 256
    rng = np.random.default_rng(5)
    n_real, cap = [7, 0, 12], 12
    prop = np.full((B, cap, 6), -1.0, np.float32)
    for b, n in enumerate(n_real):
        cx, cy = rng.uniform(20, 236, n), rng.uniform(20, 236, n)
        w, h = rng.uniform(10, 300, n), rng.uniform(10, 300, n)
        prop[b, :n] = np.stack([cx, cy, w, h, rng.integers(0, 5, n), rng.uniform(0.5, 1, n)], 1)
    fmaps = [to_half(rng.normal(size=(B, H // s, W // s, 128))) for s in (8, 16, 32)]
    rf_ref, rb_ref = O.pyramid_roi_align([f.astype(np.float64) for f in fmaps], O.mask_distribute(prop, 2, 36), (H, W), (14, 14))
    rf, rb = PyramidRoiAlign((14, 14)).crop_levels([dev(f) for f in fmaps], dev(prop), (H, W), has_k=False, base_size=36)
    np.testing.assert_array_equal(host(rb), rb_ref)                      # boxes: fp32, exact as ever
    for g, r in zip(rf, rf_ref):
        assert g.dtype == torch.float16
        g = host(g)
        np.testing.assert_array_equal((g == -1.0).all(axis=(2, 3, 4)), (r == -1.0).all(axis=(2, 3, 4)))   # MoldBatch padding
        close_half(g, r)


@pytest.mark.parametrize("cmid,K,ncls,levels,hw", [
    (128, 128, 3, [(2, 3), (2, 1), (2, 5)], (14, 14)),
    (256, 256, 3, [(1, 2), (1, 3)], (14, 14)),
    (128, 64, 5, [(2, 9), (2, 30)], (7, 7)),
    (128, 128, 3, [(8, 100)], (14, 14))])
def test_deconv_tail_half_storage(cmid, K, ncls, levels, hw):
    """csrc/deconv_out.hip with half x / half transposed-conv weights (v_mfma_f32_32x32x16_f16), fp32 1x1 conv, fp32
    sigmoid output: no rounding after the inputs, so the fp32 result is compared directly."""
    from masklab_hip import _lib, ops, packing
    h, w_ = hw
    B = levels[0][0]
    total = sum(n for _, n in levels)
    out = torch.full((B, total, 2 * h, 2 * w_, ncls), -7.0, device="cuda")
    per_roi = 4 * h * w_ * ncls
    want = np.zeros((B, total, 2 * h, 2 * w_, ncls))
    problems, off = [], 0
    for _, n in levels:
        x = to_half(rnd(B * n, h, w_, K))
        wd, bd = rnd(2, 2, cmid, K, scale=0.05), rnd(cmid)
        wo, bo = rnd(1, 1, cmid, ncls, scale=0.1), rnd(ncls)
        t = T.relu(T.conv2d_transpose_2x2_s2(x.astype(np.float64), h64(wd), bd))
        want[:, off:off + n] = T.sigmoid(T.conv2d(t, wo, bo)).reshape(B, n, 2 * h, 2 * w_, ncls)
        table, bo_p, cp = packing.pack_out1x1_table(wo, bo)
        problems.append(dict(x=dev(x), dc=ops.DeviceConv(packing.pack_transpose2x2(wd, bd), "cuda"),
                             wo_table=dev(table), bo=dev(bo_p), out=out, out_base=off * per_roi, rois_per_image=n))
        off += n
    ops.deconv2x2_out1x1_multi(problems, ncls, _lib.ACT_RELU, _lib.ACT_SIGMOID)
    np.testing.assert_allclose(host(out), want, atol=2e-5)


# ------------------------------------------------------------------ the model
def _build(bt, seed):
    from masklab_hip import ModelConfiguration, retinamasklab as R
    cfg = ModelConfiguration()
    cfg.backbone.backbone_type = bt
    _, model = R.construct_masklab_networks(cfg)
    w = model.init_weights(seed)
    model.load_weights(w, "cuda:0")
    return cfg, model, w


@pytest.mark.parametrize("bt", ["resnext50", "resnext101"])
def test_heads_run_on_half_tensors(bt):
    """No fp32 activation tensor between the stem and the predictions: every conv / GroupNorm / resize launch of the
    forward is a half one, and nothing casts the taps."""
    from masklab_hip import ops
    cfg, model, w = _build(bt, 3)
    images = np.random.default_rng(1234).integers(0, 256, (2, 256, 256, 3), dtype=np.uint8)
    ops.PROFILE = []
    got = model.predict(images)
    recs, ops.PROFILE = ops.PROFILE, None
    kernels = {r["kernel"] for r in recs}
    assert {"conv1x1_pipe_h", "gconv3x3_mfma4_h", "groupnorm_chunk_h", "resize_bilinear_h", "dwconv3x3_h"} <= kernels, kernels
    assert any(k.startswith("conv_mfma_128x") and k.endswith("_h") for k in kernels), kernels
    fp32_convs = [r["kernel"] for r in recs if r["kernel"].startswith(("conv_mfma", "conv1x1")) and not r["kernel"].endswith("_h")]
    # round 4: the stem (which reads the fp32 NHWC4 image) is fused with the max-pool into a kernel of its own, and the
    # 32-wide groups of the last stage run the half grouped kernel -- NO conv of the forward touches an fp32 activation
    assert fp32_convs == [], fp32_convs
    assert "stem7x7s2_pool_h" in kernels and "cast_h2f" not in kernels and "maxpool3x3s2_h" not in kernels, kernels
    assert "groupnorm_chunk" not in kernels and "resize_bilinear" not in kernels and "dwconv3x3" not in kernels
    want = O.inference_forward(cfg, w, images, literal_groups=False)
    for name, g, r in zip(model.output_names, got, want):
        assert g.shape == r.shape and g.dtype == np.float32, name
        if name != "roi_boxes":
            assert float(np.abs(g.astype(np.float64) - r).max()) <= F16_MODEL_TOL, name


def test_half_groupnorm_statistics_from_the_conv_epilogue():
    """Round 4: ml_conv2d_desc.gn_partials on HALF tensors (the towers / decoder of the fp16-storage mode, reference
    engine/layers/detection.py:120-125, semantic.py:205-213).  The half head conv also sums what it STORES -- the values
    after the one rounding to half, which is what this mode's GroupNorm statistics pass reads -- per 128-row tile and
    wave; GroupNorm then needs no statistics pass.  The conv's output is untouched, the pairs equal the fp64 sums of
    the stored halves, conv -> GroupNorm gives the two-pass result to within one half step, and the tower helper
    picks the form by itself (no gn_multi_stats launch for the big level)."""
    from masklab_hip import _lib, ops, packing
    from masklab_hip.keras_like import Conv2D
    from masklab_hip.layers.detection import _TowerMixin
    from masklab_hip.normalization import GroupNormalization
    B, H, W = 3, 128, 128                         # 384 tiles; chunk = 1024 pixels = 8 tiles
    x = to_half(rnd(B, H, W, 128))
    w, b = rnd(3, 3, 128, 128, scale=0.03), rnd(128)
    gamma, beta = RNG.uniform(0.5, 1.5, 128).astype(np.float32), rnd(128)
    dc = ops.DeviceConv(packing.pack_dense(w, b), "cuda")
    plain = ops.conv2d(dev(x), dc, act=_lib.ACT_RELU)
    assert plain.dtype == torch.float16
    want = host(ops.groupnorm_chunk(plain.clone(), dev(gamma), dev(beta), 16))
    part = torch.full((B * H * W // 128, 4, 2), float("nan"), dtype=torch.float64, device="cuda")
    y = ops.conv2d(dev(x), dc, act=_lib.ACT_RELU, gn_partials=part)
    np.testing.assert_array_equal(host(y), host(plain))                     # the conv's own output is untouched
    yh = host(y).astype(np.float64).reshape(-1, 128 * 128)                  # one row per 128-pixel tile
    np.testing.assert_allclose(host(part)[..., 0].sum(1), yh.sum(1), rtol=1e-6)      # (four halves are folded in fp32 first)
    np.testing.assert_allclose(host(part)[..., 1].sum(1), (yh * yh).sum(1), rtol=1e-6)
    (got,) = ops.groupnorm_chunk_multi([dict(x=y, gamma=dev(gamma), beta=dev(beta), groups=16, out=y, partials=(part, 32))])
    np.testing.assert_allclose(host(got).astype(np.float32), want.astype(np.float32), rtol=HALF_RTOL, atol=HALF_ATOL)
    with pytest.raises(RuntimeError, match="gn_partials"):                  # 64 tiles: the library would narrow / split this launch
        ops.conv2d(dev(x[:1, :64]), dc, act=_lib.ACT_RELU, gn_partials=part)
    conv, gn = Conv2D(128, (3, 3), activation='relu', padding='same', name="t/conv0"), GroupNormalization(16, name="t/gn0")
    conv.build((None, None, None, 128)); gn.build((None, None, None, 128))
    wd = {"t/conv0/kernel": w, "t/conv0/bias": b, "t/gn0/gamma": gamma, "t/gn0/beta": beta}
    conv.load_weights(wd, torch.device("cuda:0")); gn.load_weights(wd, torch.device("cuda:0"))
    xs = [to_half(rnd(B, 128, 128, 128)), to_half(rnd(B, 64, 64, 128)), to_half(rnd(B, 16, 16, 128))]
    ops.PROFILE = []
    outs = _TowerMixin._run_towers_multi([[conv, gn]] * 3, [dev(v) for v in xs])
    recs, ops.PROFILE = ops.PROFILE, None
    for v, o in zip(xs, outs):
        # two roundings between the oracle and the result (the conv's store, whose fp32 sum may round the other way in a
        # few elements, and the GroupNorm's store): two half steps
        ref = T.relu(T.conv2d(h64(v), h64(w), b.astype(np.float64))).astype(np.float16).astype(np.float64)   # the stored conv output
        np.testing.assert_allclose(host(o).astype(np.float32), T.group_norm(ref, gamma, beta, 16).astype(np.float32),
                                   rtol=2 * HALF_RTOL, atol=2e-3)


def test_resnext101_1280_half_storage_detections_match_fp32_oracle():
    """BASELINE configs[4] at its size: ResNeXt-101, one 1280x1280 image, fp16 storage end to end, against the fp32
    oracle forward.  Float outputs within 3e-2; the DETECTIONS are the oracle's (DetectionIOUMetric precision = recall =
    F = 1.0, same count per class) on a fixture whose min_confidence sits in a score gap several times wider than the
    fp16 score deviation and whose oracle detections do not change under that much score noise; roi_masks are compared
    after matching rows (row order may differ between near-equal scores in this mode).  Also: the detection stage
    itself is exact -- the oracle's DetectionProposal on the GPU's own cls_pred / loc_pred returns the GPU's rows."""
    from masklab_hip import ops
    from oracle import fixtures as FX
    from oracle import metrics as OM
    size, bt = 1280, "resnext101"
    cfg, model, w = _build(bt, 0)
    w_fix = FX.scale_cls_logits(w, FX.KNOWN_SCALE[(bt, size)])
    model.reload_class_outputs(w_fix)
    images = np.random.default_rng(1234).integers(0, 256, (1, size, size, 3), dtype=np.uint8)
    ref_cls, ref_loc = O.inference_forward(cfg, w_fix, images, literal_groups=False, with_instance=False, with_semantic=False)
    # GPU forward once with the default threshold: measures the score deviation of the mode
    got0 = dict(zip(model.output_names, model.predict(images)))
    dev_cls = float(np.abs(got0["cls_pred"].astype(np.float64) - ref_cls).max())
    assert dev_cls <= F16_MODEL_TOL
    boxes_ref = FX.boxes_from(cfg, ref_loc, size, size)
    det = cfg.detection
    args = (det.nms_iou_threshold, det.post_iou_threshold, det.nms_max_output_size)

    def stable(thr, base):              # the oracle's own detections do not change under the mode's score deviation
        rng = np.random.default_rng(0)
        for _ in range(4):
            noisy = (ref_cls.astype(np.float64) + rng.uniform(-dev_cls, dev_cls, ref_cls.shape)).astype(np.float32)
            p2, _ = O.detection_proposal(noisy, boxes_ref, thr, *args)
            if abs(float(OM.detection_iou_metric(p2, base)[2][0]) - 1.0) > 1e-6:
                return False
        return True

    # min_confidence: the widest score gaps in (0.5, 0.8), widest first, until one gives a usable fixture -- a gap wider
    # than 2.2x the deviation (the threshold sits in its middle: no score can cross it), >= 8 detections, detections
    # stable under that deviation
    sc = np.sort(ref_cls[(ref_cls > 0.5) & (ref_cls < 0.8)].astype(np.float64))
    gaps = np.diff(sc)
    thr = None
    for i in np.argsort(-gaps)[:12]:
        if gaps[i] <= 2.2 * dev_cls:
            break
        cand = float(np.float32((sc[i] + sc[i + 1]) / 2))
        base, _ = O.detection_proposal(ref_cls, boxes_ref, cand, *args)
        if (base[0, :, 4] >= 0).sum() >= 8 and stable(cand, base):
            thr = cand
            break
    assert thr is not None, ("no usable score gap", dev_cls, np.sort(gaps)[-5:])
    cfg.detection.min_confidence = thr
    model.detection_proposal.min_confidence = thr
    got = dict(zip(model.output_names, model.predict(images, want_kept=True)))
    want = dict(zip(model.output_names, O.inference_forward(cfg, w_fix, images, literal_groups=False)))
    for n in ("cls_pred", "loc_pred", "seg_pred"):
        assert float(np.abs(got[n].astype(np.float64) - want[n]).max()) <= F16_MODEL_TOL, n
    pr, rc, fm = OM.detection_iou_metric(got["roi_boxes"], want["roi_boxes"])
    assert max(abs(float(v[0]) - 1.0) for v in (pr, rc, fm)) <= 1e-6, (pr, rc, fm)
    gb, wb = got["roi_boxes"][0], want["roi_boxes"][0]
    gv, wv = gb[gb[:, 4] >= 0], wb[wb[:, 4] >= 0]
    assert len(gv) == len(wv) and sorted(gv[:, 4].tolist()) == sorted(wv[:, 4].tolist())
    # rows matched by (class, nearest centre): boxes within a pixel, masks within tolerance
    used = set()
    for i, row in enumerate(gb):
        if row[4] < 0:
            continue
        cand = [j for j in range(len(wb)) if wb[j, 4] == row[4] and j not in used]
        j = min(cand, key=lambda j: float(np.abs(wb[j, :2] - row[:2]).sum()))
        used.add(j)
        np.testing.assert_allclose(row[:4], wb[j, :4], atol=1.5)
        assert abs(float(row[5] - wb[j, 5])) <= F16_MODEL_TOL
        assert float(np.abs(got["roi_masks"][0, i] - want["roi_masks"][0, j]).max()) <= F16_MODEL_TOL
    # the detection stage on the GPU's own predictions is the oracle's, row for row
    boxes_gpu = FX.boxes_from(cfg, got["loc_pred"], size, size)
    p_self, kept_self = O.detection_proposal(got["cls_pred"], boxes_gpu, thr, *args)
    d = model.last_detections
    n = int(d["counts"].cpu()[0])
    np.testing.assert_array_equal(d["kept"].cpu().numpy()[0, :n], kept_self[:, 1:])

    # ---- BASELINE configs[4]'s WHOLE per-GPU shard: 16 x 1280 x 1280 in ONE forward (what bench.py's other_configs /
    # the *_f16 workload times); image 0 of the batch is the fixture image
    B = 16
    batch = np.random.default_rng(1234).integers(0, 256, (B, size, size, 3), dtype=np.uint8)
    assert np.array_equal(batch[:1], images)
    alone0 = got
    outs = model.predict(batch, want_kept=True)
    db = model.last_detections
    lcounts = db["level_counts"].cpu().numpy()
    counts_b, kept_b = db["counts"].cpu().numpy(), db["kept"].cpu().numpy()
    names = model.output_names
    one = dict(zip(names, FX.image_of_batch(names, outs, lcounts, 0)))
    # image 0 of the batch vs the fp32 oracle: the same bars as the image alone
    for n_ in ("cls_pred", "loc_pred", "seg_pred"):
        assert float(np.abs(one[n_].astype(np.float64) - want[n_]).max()) <= F16_MODEL_TOL, n_
    pr, rc, fm = OM.detection_iou_metric(one["roi_boxes"], want["roi_boxes"])
    assert max(abs(float(v[0]) - 1.0) for v in (pr, rc, fm)) <= 1e-6, (pr, rc, fm)
    # image 0 of the batch vs image 0 alone: launches that are cut along K for one image and not for sixteen sum K in
    # other pieces, which moves half roundings of the stored activations -- a tenth of the mode's bar
    for n_ in ("cls_pred", "loc_pred", "seg_pred"):
        dev_ = float(np.abs(one[n_] - alone0[n_]).max())
        print(f"f16s batch-16 vs alone, {n_}: {dev_:.3e}")
        assert dev_ <= 0.1 * F16_MODEL_TOL, (n_, dev_)
    assert int(counts_b[0]) == n
    assert sorted(map(tuple, kept_b[0, :n].tolist())) == sorted(map(tuple, kept_self[:, 1:].tolist()))
    # the detection stage is exact for the images of the batch too: the oracle's DetectionProposal on the GPU's own
    # predictions of image k returns the GPU's rows
    for k in (0, 9, 15):
        bk = FX.boxes_from(cfg, outs[names.index("loc_pred")][k:k + 1], size, size)
        _, kept_k = O.detection_proposal(outs[names.index("cls_pred")][k:k + 1], bk, thr, *args)
        np.testing.assert_array_equal(kept_b[k, :counts_b[k]], kept_k[:, 1:])
    ops.set_conv_math("f16s")
