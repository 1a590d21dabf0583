"""GPU parity of the fp16 MFMA path (BASELINE config 5: ResNeXt-101, fp16 MFMA): `ops.set_conv_math("f16")`
makes the dense convolutions round their operands to fp16 and accumulate in fp32 on
v_mfma_f32_32x32x16_f16; tensors in HBM stay fp32.

Kernel-level bar: against the oracle conv evaluated on the SAME fp16-rounded operands (fp64 accumulation)
the result agrees to fp32-accumulation accuracy (atol 2e-5 on O(1) data) -- i.e. the only difference to the
fp32 path is the documented operand rounding.  Model-level bar: against the fp32 oracle forward the fp16
path's outputs agree within 3e-2 (operand rounding 2^-11 per layer through ~100 layers); index outputs are
not required to be bit-exact in this mode (a score within that distance of a threshold may flip).  -m gpu."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import masklab as O
from oracle import tfops as T

RNG = np.random.default_rng(23)
F16_MODEL_TOL = 3e-2


def rnd(*shape, scale=1.0):
    return (RNG.normal(size=shape) * scale).astype(np.float32)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    torch.cuda.synchronize()
    return t.cpu().numpy()


def h(a):
    """round to fp16 (RNE) and back: what the kernel does to an operand on its way into LDS"""
    return a.astype(np.float16).astype(np.float64)


@pytest.fixture(scope="module", autouse=True)
def _f16_mode():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from masklab_hip import ops
    ops.set_conv_math("f16")
    yield
    ops.set_conv_math("f32")


@pytest.mark.parametrize("k,cin,cout,stride,padding,dil,act,hw", [
    (1, 64, 256, 1, "valid", 1, "relu", (40, 24)),        # 128x128 tile
    (3, 128, 128, 1, "same", 1, "relu", (33, 35)),        # halo taps, ragged M
    (3, 128, 75, 1, "same", 1, "sigmoid", (16, 16)),      # 128x32 tile
    (3, 128, 60, 1, "same", 1, None, (8, 8)),             # 128x64 tile
    (3, 2048, 128, 2, "same", 1, "relu", (4, 4)),         # long K, stride 2
    (1, 8, 128, 1, "valid", 1, "sigmoid", (1, 1)),        # cin < 32 (zero padded K)
    (3, 32, 96, 1, "same", 3, "relu6", (12, 12)),         # dilation
])
def test_conv2d_f16_math(k, cin, cout, stride, padding, dil, act, hw):
    from masklab_hip import _lib, ops, packing
    x = rnd(2, hw[0], hw[1], cin)
    w, b = rnd(k, k, cin, cout, scale=1.0 / np.sqrt(k * k * cin)), rnd(cout)
    ref = T.conv2d(h(x), h(w), b.astype(np.float64), stride, padding, dil)      # bias is added in fp32, unrounded
    ref = {"relu": T.relu, "relu6": T.relu6, "sigmoid": T.sigmoid, None: lambda v: v}[act](ref)
    got = host(ops.conv2d(dev(x), ops.DeviceConv(packing.pack_dense(w, b), "cuda"), stride=stride, padding=padding,
                          dilation=dil, act=_lib.ACT_BY_NAME[act]))
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-5)
    # and it really is the fp16 path: the unrounded fp32 reference differs by more than that
    if k * k * cin >= 64:
        exact = T.conv2d(x.astype(np.float64), w, b, stride, padding, dil)
        exact = {"relu": T.relu, "relu6": T.relu6, "sigmoid": T.sigmoid, None: lambda v: v}[act](exact)
        assert np.abs(got - exact).max() > 2e-5


def test_conv2d_f16_split_k_and_multi_problem():
    from masklab_hip import _lib, ops, packing
    x = rnd(2, 6, 6, 2048)
    w, b = rnd(3, 3, 2048, 128, scale=0.01), rnd(128)
    ref = T.relu(T.conv2d(h(x), h(w), b.astype(np.float64), 2, "same"))
    got = host(ops.conv2d(dev(x), ops.DeviceConv(packing.pack_dense(w, b), "cuda"), stride=2, padding="same",
                          act=_lib.ACT_RELU))
    np.testing.assert_allclose(got, ref, atol=5e-5)
    xs = [rnd(1, s, s, 128) for s in (16, 8, 4)]
    ws = [(rnd(3, 3, 128, 128, scale=0.03), rnd(128)) for _ in xs]
    outs = ops.conv2d_multi([dict(x=dev(xi), dc=ops.DeviceConv(packing.pack_dense(wi, bi), "cuda"), act=_lib.ACT_RELU)
                             for xi, (wi, bi) in zip(xs, ws)])
    for xi, (wi, bi), o in zip(xs, ws, outs):
        np.testing.assert_allclose(host(o), T.relu(T.conv2d(h(xi), h(wi), bi.astype(np.float64))), atol=2e-5)


def test_conv2d_f16_stem_residual_and_transpose():
    from masklab_hip import _lib, ops, packing
    # NHWC4 row-span stem
    x = rnd(1, 37, 41, 3)
    x4 = np.concatenate([x, np.zeros_like(x[..., :1])], -1)
    w, b = rnd(7, 7, 3, 64, scale=0.08), rnd(64)
    ref = T.relu(T.conv2d(h(x), h(w), b.astype(np.float64), 2, ((3, 3), (3, 3))))
    got = host(ops.conv2d(dev(x4), ops.DeviceConv(packing.pack_rowspan(w, b), "cuda"), stride=2,
                          padding=((3, 3), (3, 3)), act=_lib.ACT_RELU))
    np.testing.assert_allclose(got, ref, atol=2e-5)
    # residual add stays fp32
    x, res = rnd(2, 12, 12, 64), rnd(2, 12, 12, 96)
    w, b = rnd(1, 1, 64, 96, scale=0.1), rnd(96)
    ref = T.relu(T.conv2d(h(x), h(w), b.astype(np.float64), padding="valid") + res)
    got = host(ops.conv2d(dev(x), ops.DeviceConv(packing.pack_dense(w, b), "cuda"), padding="valid",
                          act=_lib.ACT_RELU, residual=dev(res)))
    np.testing.assert_allclose(got, ref, atol=2e-5)


@pytest.mark.parametrize("bt", ["resnext101", "resnext50"])
def test_full_forward_f16_close_to_fp32_oracle(bt):
    from masklab_hip import ModelConfiguration, retinamasklab as R
    cfg = ModelConfiguration()
    cfg.backbone.backbone_type = bt
    _, model = R.construct_masklab_networks(cfg)
    w = model.init_weights(3)
    model.load_weights(w, "cuda:0")
    images = np.random.default_rng(1234).integers(0, 256, (2, 128, 128, 3), dtype=np.uint8)
    got = model.predict(images)
    want = O.inference_forward(cfg, w, images, literal_groups=False)
    worst = {}
    for name, g, r in zip(model.output_names, got, want):
        assert g.shape == r.shape, name
        if name == "roi_boxes":
            continue                                    # no detections at the reference init (scores ~0.01)
        worst[name] = float(np.abs(g.astype(np.float64) - r).max())
        assert worst[name] <= F16_MODEL_TOL, (name, worst[name])
    assert max(worst.values()) > 1e-6, "suspiciously exact: the fp16 path did not run"
