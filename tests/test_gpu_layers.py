"""Layer-level GPU parity (SURVEY 8a rows that round 1 covered only end to end): ASPPNetwork / AtrousSeparableConv2D
(a15), SegmentationSubNet (a16), FeaturePyramid (a7) and the P6 / P7 extra levels of load_backbone (a2), each layer
object of masklab_hip against the oracle function restated from the reference, on its own inputs.  -m gpu."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import masklab as O
from oracle import tfops as T

RNG = np.random.default_rng(41)
TOL = 1e-4


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


@pytest.fixture(autouse=True)
def _fresh_layer_names():
    """Default layer names carry a per-session counter (`segmentation_sub_net_1` after another test built one):
    the oracle functions look the weights up under the un-numbered default."""
    from masklab_hip import keras_like as K
    K.clear_session()


def _loaded(layer, shape, seed=0):
    from masklab_hip import keras_like as K
    layer.build(shape)
    w = K.init_weights(layer.weight_specs(), seed)
    layer.load_weights(w, torch.device("cuda:0"))
    return w


@pytest.mark.parametrize("hw,cin", [((32, 32), 256), ((20, 28), 512)])
def test_aspp_network_layer(hw, cin):
    """reference engine/layers/semantic.py:93-168 (+ AtrousSeparableConv2D :32-90): dilations 6/12/18 on a map where
    the dilated taps leave the image, the ReLU-without-GN pooling branch, the 5-way concat projection"""
    from masklab_hip.layers import ASPPNetwork
    layer = ASPPNetwork(num_features=128, atrous_rate=(6, 12, 18), groups=16)
    w = _loaded(layer, (None, hw[0], hw[1], cin))
    x = np.maximum(rnd(2, hw[0], hw[1], cin), 0)
    got = host(layer(dev(x)))
    want = O.aspp_network(x.astype(np.float64), w, (6, 12, 18), 16)
    assert got.shape == want.shape == (2, hw[0], hw[1], 128)
    np.testing.assert_allclose(got, want, atol=TOL)


def test_atrous_separable_conv2d_layer():
    from masklab_hip.layers import AtrousSeparableConv2D
    layer = AtrousSeparableConv2D(128, dilation_rate=12, groups=16, name="aspp_12")
    w = _loaded(layer, (None, 16, 24, 256), seed=3)
    x = rnd(1, 16, 24, 256)
    y = T.depthwise_conv2d(x.astype(np.float64), w["aspp_12_depthwise/depthwise_kernel"], dilation=12)
    y = T.relu(T.group_norm(y, w["aspp_12_depthwise_GN/gamma"], w["aspp_12_depthwise_GN/beta"], 16))
    y = T.conv2d(y, w["aspp_12_pointwise/kernel"])
    y = T.relu(T.group_norm(y, w["aspp_12_pointwise_GN/gamma"], w["aspp_12_pointwise_GN/beta"], 16))
    np.testing.assert_allclose(host(layer(dev(x))), y, atol=TOL)


def test_segmentation_subnet_layer():
    """semantic.py:178-246: skip projection (C/G = 2), align-corners upsample of the ASPP map into the concat buffer,
    the conv->ReLU->GN tower on 160 channels, sigmoid 1x1"""
    from masklab_hip.layers import SegmentationSubNet
    layer = SegmentationSubNet(num_depth=2, num_features=128, num_skip_features=32, num_classes=3, groups=16)
    w = _loaded(layer, [(None, 8, 12, 128), (None, 32, 48, 256)], seed=5)
    aspp, skip = rnd(2, 8, 12, 128), np.maximum(rnd(2, 32, 48, 256), 0)
    got = host(layer([dev(aspp), dev(skip)]))
    want = O.segmentation_subnet(aspp.astype(np.float64), skip.astype(np.float64), w, 2, 16)
    assert got.shape == want.shape == (2, 32, 48, 3)
    np.testing.assert_allclose(got, want, atol=TOL)


def test_feature_pyramid_layer():
    """detection.py:30-74: laterals, align-corners top-down merge of the PRE-3x3 sum, un-activated P3..P5 convs,
    on level sizes that are not exact halves of each other"""
    from masklab_hip.layers import FeaturePyramid
    layer = FeaturePyramid(strides=[8, 16, 32], num_features=128)
    shapes = [(None, 25, 37, 64), (None, 13, 19, 96), (None, 7, 10, 160)]
    w = _loaded(layer, shapes, seed=7)
    xs = [rnd(2, *s[1:]) for s in shapes]
    got = [host(t) for t in layer([dev(x) for x in xs])]
    want = O.feature_pyramid([x.astype(np.float64) for x in xs], w, [8, 16, 32])
    for g, r in zip(got, want):
        assert g.shape == r.shape
        np.testing.assert_allclose(g, r, atol=TOL)


@pytest.mark.parametrize("bt", ["resnext50", "mobilenet"])
def test_extra_levels_p6_p7(bt):
    """engine/backbone/base.py:285-316: P6 = ReLU(conv 3x3 s2 on C5) exported PRE-norm, P6_norm with the default 32
    groups, P7 on the normalised tensor; 'same' padding (ResNeXt) vs ZeroPadding2D(((0,1),(0,1))) + valid (MobileNet)
    on odd map sizes"""
    from masklab_hip import backbone as BB
    from masklab_hip import keras_like as K
    K.clear_session()
    bb = BB.load_backbone(bt, backbone_outputs=("C5", "P6", "P7"), num_features=128)
    w = K.init_weights(bb.weight_specs(), 1)
    bb.load_weights(w, torch.device("cuda:0"))
    images = np.random.default_rng(3).integers(0, 256, (2, 160, 224, 3), dtype=np.uint8)
    got = [host(t) for t in bb(dev(images))]
    names, want = O.backbone_forward(images.astype(np.float32), w, bt, ("C5", "P6", "P7"), literal_groups=False)
    assert names == bb.output_names == ["C5", "P6", "P7"]
    for n, g, r in zip(names, got, want):
        assert g.shape == r.shape, n
        assert float(np.max(np.abs(g.astype(np.float64) - r))) <= 1e-3, n
