"""GPU parity of the fp16 MFMA path WITH fp16 STORAGE (BASELINE config 5; `ops.set_conv_math("f16s")`): the ResNeXt
body keeps activations and weights as IEEE half in HBM -- stem (fp32 image -> half), max-pool, bottleneck 1x1 convs on
the persistent pipelined kernel (csrc/conv1x1_pipe.hip, _Float16 instantiation), grouped 3x3, strided shortcuts --
and hands HALF taps to the heads (their kernels: tests/test_gpu_f16_heads.py).

Kernel-level bar: against the oracle op evaluated on the SAME half-rounded operands in fp64, with the SAME single
rounding of the result to half: equal to within one half ulp-step (a result within fp32-accumulation distance of a
rounding boundary may land on the neighbouring half value): rtol 2^-10, atol 1e-4.  Byte-moving kernels: exact.
Model-level bar: against the fp32 oracle forward within 3e-2, as for the fp16-operand mode (tests/test_gpu_f16.py).
-m gpu."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import masklab as O
from oracle import tfops as T

RNG = np.random.default_rng(29)
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


@pytest.fixture(scope="module", autouse=True)
def _f16s_mode():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from masklab_hip import ops
    ops.set_conv_math("f16s")
    yield
    ops.set_conv_math("f32")


ACT = {"relu": T.relu, "relu6": T.relu6, None: lambda v: v}


@pytest.mark.parametrize("cin,cout,hw,B,res,act,stride", [
    (64, 128, (40, 24), 2, False, "relu", 1),        # K = 64 halves: ONE chunk per tile (every chunk is a first chunk)
    (64, 256, (37, 29), 3, False, "relu", 1),        # ragged M, two N tiles
    (128, 256, (32, 32), 2, True, "relu", 1),        # bottleneck exit: half residual + ReLU
    (256, 128, (33, 31), 2, False, None, 1),         # linear
    (256, 512, (16, 16), 2, True, "relu6", 1),       # N tiles split over work units
    (1024, 2048, (8, 8), 1, True, "relu", 1),        # 16 N tiles in groups (bias slice per block), K = 1024
    (256, 512, (32, 30), 2, False, None, 2),         # strided shortcut: sampled, then the stride-1 kernel
    (192, 128, (20, 20), 1, False, "relu", 1),       # K = 192: three chunks
])
def test_conv1x1_half_storage(cin, cout, hw, B, res, act, stride):
    from masklab_hip import _lib, ops, packing
    x = to_half(rnd(B, hw[0], hw[1], cin))
    w, b = rnd(1, 1, cin, cout, scale=1.0 / np.sqrt(cin)), rnd(cout)
    Ho, Wo = (hw[0] + stride - 1) // stride, (hw[1] + stride - 1) // stride
    r = to_half(rnd(B, Ho, Wo, cout)) if res else None
    ref = T.conv2d(x.astype(np.float64), h64(w), b.astype(np.float64), stride, "valid", 1)
    if res:
        ref = ref + r.astype(np.float64)
    ref = ACT[act](ref).astype(np.float16)
    dc = ops.DeviceConv(packing.pack_dense(w, b), "cuda")
    got = ops.conv2d(dev(x), dc, stride=stride, padding="valid", act=_lib.ACT_BY_NAME[act],
                     residual=dev(r) if res else None)
    assert got.dtype == torch.float16 and tuple(got.shape) == ref.shape
    np.testing.assert_allclose(host(got).astype(np.float32), ref.astype(np.float32), rtol=HALF_RTOL, atol=HALF_ATOL)


@pytest.mark.parametrize("c,C,stride,hw", [(4, 128, 1, (24, 20)), (8, 256, 2, (20, 24)), (16, 512, 1, (16, 16)),
                                           (4, 128, 2, (17, 19)), (8, 256, 1, (9, 30)), (16, 512, 2, (19, 21)),
                                           (16, 1024, 1, (5, 7)), (32, 1024, 1, (40, 40)), (32, 1024, 2, (21, 19)),
                                           (32, 128, 1, (9, 13))])
def test_grouped3x3_half_storage(c, C, stride, hw):
    """fp16 tensors run the grouped 3x3 on the fp16 matrix instructions (v_mfma_f32_4x4x4_16B_f16 for c = 4 / 8,
    v_mfma_f32_16x16x16_f16 for c = 16, v_mfma_f32_32x32x16_f16 for c = 32): activations are the tensor's own halves, the kernel is rounded to half once per
    block, products exact, fp32 accumulation -- so the oracle gets BOTH operands half-rounded"""
    from masklab_hip import _lib, ops, packing
    groups = C // c
    x = to_half(rnd(2, hw[0], hw[1], C))
    k, b = rnd(3, 3, C, c, scale=1.0 / np.sqrt(9 * c)), rnd(C)
    ref = O.grouped_conv_fast(x.astype(np.float64), h64(k), groups, c, stride) + b.astype(np.float64)
    ref = T.relu(ref).astype(np.float16)
    got = ops.gconv3x3(dev(x), dev(packing.pack_grouped_mfma4(k, groups)), dev(b), c, stride=stride,
                       padding=((1, 1), (1, 1)), act=_lib.ACT_RELU)
    assert got.dtype == torch.float16
    np.testing.assert_allclose(host(got).astype(np.float32), ref.astype(np.float32), rtol=HALF_RTOL, atol=HALF_ATOL)


def test_half_byte_movers_are_exact():
    from masklab_hip import ops
    x = to_half(np.abs(rnd(2, 21, 18, 64)))               # post-ReLU map
    ref = T.max_pool(np.pad(x.astype(np.float64), ((0, 0), (1, 1), (1, 1), (0, 0))), 3, 2)
    np.testing.assert_array_equal(host(ops.maxpool3x3s2(dev(x), pad=1)), ref.astype(np.float16))
    y = to_half(rnd(3, 9, 12, 32))
    np.testing.assert_array_equal(host(ops.subsample2_h(dev(y))), y[:, ::2, ::2])
    np.testing.assert_array_equal(host(ops.cast_h2f(dev(y))), y.astype(np.float32))


@pytest.mark.parametrize("B,H,W", [(2, 64, 96), (1, 37, 41), (3, 130, 70)])
def test_fused_stem_and_pool_equals_the_two_kernels(B, H, W):
    """csrc/stem_h.hip (round 4): the fp16-storage stem -- 7x7 stride-2 conv + folded BN + ReLU -- and the 3x3 stride-2
    max-pool in ONE kernel (reference engine/backbone/ResNext.py:343-352).  Same operands (image and weights rounded to half),
    same k order, bias first, one rounding, exact max: BIT-identical to the unfused pair; and within one half step of the
    oracle.  Odd sizes exercise the zero padding on every side and partial pooled tiles."""
    from masklab_hip import _lib, ops, packing
    x = rnd(B, H, W, 3)
    x4 = np.concatenate([x, np.zeros_like(x[..., :1])], -1)
    w, b = rnd(7, 7, 3, 64, scale=0.08), rnd(64)
    dc = ops.DeviceConv(packing.pack_rowspan(w, b), "cuda")
    two = ops.maxpool3x3s2(ops.conv2d(dev(x4), dc, stride=2, padding=((3, 3), (3, 3)), act=_lib.ACT_RELU, out_dtype=torch.float16), pad=1)
    one = ops.stem_pool_h(dev(x4), dc)
    assert one.dtype == torch.float16 and one.shape == two.shape
    np.testing.assert_array_equal(host(one), host(two))
    conv = T.relu(T.conv2d(h64(x), h64(w), b.astype(np.float64), 2, ((3, 3), (3, 3)))).astype(np.float16).astype(np.float64)
    ref = T.max_pool(np.pad(conv, ((0, 0), (1, 1), (1, 1), (0, 0))), 3, 2)
    np.testing.assert_allclose(host(one).astype(np.float32), ref.astype(np.float32), rtol=HALF_RTOL, atol=HALF_ATOL)


def test_stem_writes_half():
    """the 7x7 stride-2 stem (fp32 NHWC4 image in, operands rounded to half in the kernel) storing half"""
    from masklab_hip import _lib, ops, packing
    x = rnd(1, 37, 41, 3)
    x4 = np.concatenate([x, np.zeros_like(x[..., :1])], -1)
    w, b = rnd(7, 7, 3, 64, scale=0.08), rnd(64)
    ref = T.relu(T.conv2d(h64(x), h64(w), b.astype(np.float64), 2, ((3, 3), (3, 3)))).astype(np.float16)
    got = ops.conv2d(dev(x4), ops.DeviceConv(packing.pack_rowspan(w, b), "cuda"), stride=2, padding=((3, 3), (3, 3)),
                     act=_lib.ACT_RELU, out_dtype=torch.float16)
    assert got.dtype == torch.float16
    np.testing.assert_allclose(host(got).astype(np.float32), ref.astype(np.float32), rtol=HALF_RTOL, atol=HALF_ATOL)


def test_last_stage_groups_of_32_through_fp32_copy():
    """c = 32 groups (1024 filters): half in -> fp32 copy -> dense grouped kernel (fp16 operands) -> half out"""
    from masklab_hip import keras_like as K
    layer = K.GroupedConv2D(1024, 32, activation='relu', name="g32")
    layer.build((None, 8, 8, 1024))
    w = K.init_weights(layer.weight_specs(), 0)
    layer.load_weights(w, torch.device("cuda:0"))
    x = to_half(rnd(2, 8, 8, 1024))
    ref = T.relu(O.grouped_conv_fast(x.astype(np.float64), h64(w["g32/depthwise_kernel"]), 32, 32, 1)).astype(np.float16)
    got = layer(dev(x))
    assert got.dtype == torch.float16
    np.testing.assert_allclose(host(got).astype(np.float32), ref.astype(np.float32), rtol=HALF_RTOL, atol=2e-4)


@pytest.mark.parametrize("bt", ["resnext101", "resnext50"])
def test_full_forward_half_storage_close_to_fp32_oracle(bt):
    from masklab_hip import ModelConfiguration, ops, retinamasklab as R
    cfg = ModelConfiguration()
    cfg.backbone.backbone_type = bt
    _, model = R.construct_masklab_networks(cfg)
    w = model.init_weights(3)
    model.load_weights(w, "cuda:0")
    images = np.random.default_rng(1234).integers(0, 256, (2, 128, 128, 3), dtype=np.uint8)
    ops.PROFILE = []
    got = model.predict(images)
    recs, ops.PROFILE = ops.PROFILE, None
    kernels = {r["kernel"] for r in recs}
    assert {"conv1x1_pipe_h", "gconv3x3_mfma4_h", "stem7x7s2_pool_h"} <= kernels, kernels   # the half path ran
    want = O.inference_forward(cfg, w, images, literal_groups=False)
    worst = {}
    for name, g, r in zip(model.output_names, got, want):
        assert g.shape == r.shape and g.dtype == np.float32, name
        if name == "roi_boxes":
            continue                                    # no detections at the reference init (scores ~0.01)
        worst[name] = float(np.abs(g.astype(np.float64) - r).max())
        assert worst[name] <= F16_MODEL_TOL, (name, worst[name])
    assert max(worst.values()) > 1e-6, "suspiciously exact: the fp16 path did not run"


@pytest.mark.parametrize("cin,cout,hw,B,res,act", [
    (1024, 512, (80, 80), 11, False, "relu"),        # ResNeXt-101 stage 3 conv1: 275 panels x 2 N tiles (XCD map), 16 chunks
    (512, 1024, (80, 80), 11, True, "relu"),         # conv3 + half residual: 4 N tiles per panel
    (256, 512, (67, 61), 9, True, "relu6"),          # ragged M (36 783 rows = 143.7 panels), 4 chunks
    (512, 256, (96, 96), 8, False, None),            # one N tile per panel, no clamp
    (1024, 2048, (40, 40), 16, True, "relu"),        # 8 N tiles per panel
])
def test_conv1x1_h256_half_storage(cin, cout, hw, B, res, act):
    """csrc/conv1x1_h256.hip (half tensors, 256 x 256 tiles, the K >= 256 bottleneck convs of BASELINE configs[4]):
    against the oracle on identically rounded operands, AND bit for bit against conv1x1_pipe_kernel<_Float16> (the same
    k-ordered fp32 chains, residual then bias, one rounding)."""
    from masklab_hip import _lib, ops, packing
    lib = _lib.load()
    x = to_half(rnd(B, hw[0], hw[1], cin))
    w, b = rnd(1, 1, cin, cout, scale=1.0 / np.sqrt(cin)), rnd(cout)
    r = to_half(rnd(B, hw[0], hw[1], cout)) if res else None
    outs = {}
    for tile in (5, 4, 0):
        dc = ops.DeviceConv(packing.pack_dense(w, b, tile=tile), "cuda")
        outs[tile] = host(ops.conv2d(dev(x), dc, padding="valid", act=_lib.ACT_BY_NAME[act], residual=dev(r) if res else None))
    np.testing.assert_array_equal(outs[5], outs[4])
    np.testing.assert_array_equal(outs[0], outs[5])            # (what the automatic choice runs gives the same bits)
    sub = slice(0, 2)                                           # oracle on the first two images (it is a CPU conv)
    ref = T.conv2d(x[sub].astype(np.float64), h64(w), b.astype(np.float64), 1, "valid", 1)
    if res:
        ref = ref + r[sub].astype(np.float64)
    ref = ACT[act](ref).astype(np.float16)
    np.testing.assert_allclose(outs[5][sub].astype(np.float32), ref.astype(np.float32), rtol=HALF_RTOL, atol=HALF_ATOL)
    last = slice(B - 1, B)                                      # ... and on the last one (the ragged tail panel)
    ref = T.conv2d(x[last].astype(np.float64), h64(w), b.astype(np.float64), 1, "valid", 1)
    if res:
        ref = ref + r[last].astype(np.float64)
    ref = ACT[act](ref).astype(np.float16)
    np.testing.assert_allclose(outs[5][last].astype(np.float32), ref.astype(np.float32), rtol=HALF_RTOL, atol=HALF_ATOL)


@pytest.mark.parametrize("rows_per_block,ragged", [(512 + 32, 0), (512 + 64, 0), (512 + 96, 5), (512 + 128, 0), (512 + 160, 0),
                                                    (512 + 224, 19), (256 + 32, 0)])
def test_conv1x1_h256_row_ranges_short_tiles_and_repeats(rows_per_block, ragged):
    """Round 4 form of csrc/conv1x1_h256.hip: every block walks ONE contiguous range of rows (all ranges equal +- 32
    rows) through a ring of 3 activation + 2 weight LDS slots with counted waits, and a short last tile of up to 128
    rows takes the light path (Q = 1..4 row groups of 32), longer ones the full stream on zero-filled rows.  M is chosen
    so that each of the 128 ranges of an N = 512 launch (2 N tiles, 128 blocks each on a 256-CU part) ends in the given
    tail; `ragged` rows are cut off the end (the last range's tail is then not a multiple of 32).  Bit for bit against
    conv1x1_pipe_kernel<_Float16>, five launches each: a staged slot read before its request has landed shows up as RARE
    wrong tiles, not as a consistent error."""
    from masklab_hip import _lib, ops, packing
    cin, cout = 256, 512                         # 4 chunks per tile (the shortest K the kernel takes), 2 N tiles
    M = 128 * rows_per_block - ragged
    x = to_half(rnd(1, 1, M, cin))
    w, b = rnd(1, 1, cin, cout, scale=1.0 / np.sqrt(cin)), rnd(cout)
    r = to_half(rnd(1, 1, M, cout))
    xd, rd = dev(x), dev(r)
    ref = ops.conv2d(xd, ops.DeviceConv(packing.pack_dense(w, b, tile=4), "cuda"), padding="valid", act=_lib.ACT_RELU, residual=rd)
    dc5 = ops.DeviceConv(packing.pack_dense(w, b, tile=5), "cuda")
    for rep in range(5):
        got = ops.conv2d(xd, dc5, padding="valid", act=_lib.ACT_RELU, residual=rd)
        assert torch.equal(got, ref), f"launch {rep}: {int((got != ref).sum())} elements differ"
    sub = slice(M - 700, M)                      # the oracle on the last rows (short tile + ragged end)
    want = T.conv2d(x[:, :, sub].astype(np.float64), h64(w), b.astype(np.float64), 1, "valid", 1) + r[:, :, sub].astype(np.float64)
    np.testing.assert_allclose(host(ref)[:, :, sub].astype(np.float32), np.maximum(want, 0).astype(np.float16).astype(np.float32),
                               rtol=HALF_RTOL, atol=HALF_ATOL)
