"""GPU parity of the serving graph's arithmetic layers (reference
road_project/setup/serving.py:28-50, engine/layers/misc.py:358-401, 524-727) against the CPU oracle.
CropAndPadMask is exact (same fp32 bilinear expression); the road-width regression is float32 normal
equations in the reference (cond ~1e6), so sizes are compared with rtol 2e-3.  -m gpu."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import masklab as O


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def host(t):
    torch.cuda.synchronize()
    return t.cpu().numpy()


def _scene(B=2, n=7, H=90, W=160, seed=0, crack=True):
    rng = np.random.default_rng(seed)
    det = np.full((B, n, 6), -1, np.int32)
    ins = np.zeros((B, n, 28, 28), np.int32)
    for b in range(B):
        k = n - b                                           # image 1 has one padded row
        det[b, :k, 0] = rng.integers(-5, W + 5, k)          # boxes may overhang the canvas
        det[b, :k, 1] = rng.integers(-5, H + 5, k)
        det[b, :k, 2] = rng.integers(1, W // 2, k)
        det[b, :k, 3] = rng.integers(1, H // 2, k)
        det[b, :3, 0] = W // 2 + rng.integers(-10, 10, 3)   # three instances sit on the road
        det[b, :3, 1] = int(H * 0.7) + rng.integers(-5, 5, 3)
        det[b, :k, 4] = rng.integers(0, 5, k)
        det[b, :k, 5] = rng.integers(30, 100, k)            # some below, some above the conf 50 cut
        ins[b, :k] = (rng.random((k, 28, 28)) > 0.4).astype(np.int32)
        det[b, k:] = np.array([-2, -2, -2, -2, -1, -100])   # what UpSampleOutput makes of -1 padding
    seg = np.zeros((B, H, W, 3), np.int32)
    for b in range(B):                                      # a road trapezoid with ragged edges
        for y in range(20, H - 5):
            left = int(W * 0.45 - (y - 20) * 0.5 + rng.integers(-2, 3))
            right = int(W * 0.55 + (y - 20) * 0.6 + rng.integers(-2, 3))
            seg[b, y, max(left, 0):min(right, W), 1] = 1
        seg[b, 5:9, 100:140, 1] = 1                         # rows 9..19 have no road pixel at all (empty segments)
        seg[b, 12, 50, 1] = 1                               # a single-pixel row (min == max: dropped)
    seg[..., 0] = (rng.random((B, H, W)) > 0.7).astype(np.int32)
    if crack:
        seg[0, 40:48, 30:90, 2] = 1
        seg[1, 60:62, 100:120, 2] = 1
    return det, ins, seg


def test_crop_and_pad_mask():
    from masklab_hip.layers import CropAndPadMask
    det, ins, seg = _scene()
    H, W = seg.shape[1:3]
    want = O.crop_and_pad_mask((H, W), det, ins)
    got = host(CropAndPadMask()([dev(np.zeros((2, H, W, 3), np.uint8)), dev(det), dev(ins), dev(seg)]))
    assert got.shape == want.shape and want.max() > 0
    np.testing.assert_array_equal(got, want)
    # all confidences <= 50: threshold -100 keeps every row, padded ones included
    det2 = det.copy()
    det2[..., 5] = np.minimum(det2[..., 5], 40)
    want2 = O.crop_and_pad_mask((H, W), det2, ins)
    got2 = host(CropAndPadMask()([dev(np.zeros((2, H, W, 3), np.uint8)), dev(det2), dev(ins)]))
    np.testing.assert_array_equal(got2, want2)


@pytest.mark.parametrize("crack", [True, False])
def test_summary_output(crack):
    from masklab_hip.layers import CropAndPadMask, SummaryOutput
    det, ins, seg = _scene(seed=3, crack=crack)
    H, W = seg.shape[1:3]
    masks = O.crop_and_pad_mask((H, W), det, ins)
    want = O.summary_output(det, seg, masks, 3.25)
    got = host(SummaryOutput(3.25)([dev(det), dev(seg), dev(masks)]))
    assert got.shape == want.shape == (2, det.shape[1] + (1 if crack else 0), 11)
    np.testing.assert_array_equal(got[..., :6], want[..., :6])                   # class, box, conf (+ crack instance)
    np.testing.assert_allclose(got[..., 6], want[..., 6], rtol=1e-6)             # pixel counts
    np.testing.assert_allclose(got[..., 7:10], want[..., 7:10], rtol=2e-3)       # sizes (ill-conditioned float32 regression)
    np.testing.assert_array_equal(got[..., 10], want[..., 10])                   # include_my_road
    assert want[..., 7].max() > 0 and 0 < want[..., 10].mean() < 1


@pytest.mark.parametrize("seed,shape,crack", [(3, (2, 7, 90, 160), True), (11, (3, 9, 130, 200), False),
                                              (5, (2, 5, 64, 300), True)])
def test_summary_from_rois_is_bit_identical_to_the_materialised_path(seed, shape, crack):
    """SummaryOutput(from_rois=True) (ml_instance_summary_rois_f32: CropAndPadMask folded into the summary kernels, only
    the rows / column stripes a box covers are visited) against CropAndPadMask -> SummaryOutput on the [B,n,H,W] canvases:
    every number equal bit for bit -- boxes overhanging the canvas, rows below the confidence cut, -1 padded rows, a
    zero-sized box, boxes narrower than a 64-lane stripe and wider than 256 columns."""
    from masklab_hip.layers import CropAndPadMask, SummaryOutput
    B, n, H, W = shape
    det, ins, seg = _scene(B=B, n=n, H=H, W=W, seed=seed, crack=crack)
    det[0, 1, 2:4] = (W + 40, H + 20)                       # wider / taller than the canvas
    det[0, 2, 0:4] = (-30, -30, 4, 4)                       # clipped to nothing
    det[0, 3, 2:4] = (3, 2)                                 # a few pixels
    images = dev(np.zeros((B, H, W, 3), np.uint8))
    d, i, sg = dev(det), dev(ins), dev(seg)
    layer = SummaryOutput(3.25)
    want = host(layer([d, sg, CropAndPadMask()([images, d, i, sg])]))
    got = host(layer([d, sg, i], from_rois=True))
    assert got.shape == want.shape and want[..., 6].max() > 0
    np.testing.assert_array_equal(got, want)
    # all confidences <= 50: every row pasted (threshold -100)
    det2 = det.copy()
    det2[..., 5] = np.minimum(det2[..., 5], 40)
    d2 = dev(det2)
    np.testing.assert_array_equal(host(layer([d2, sg, i], from_rois=True)),
                                  host(layer([d2, sg, CropAndPadMask()([images, d2, i, sg])])))


def test_instance_size_without_any_road():
    from masklab_hip.layers import CalculateInstanceSize, IncludeMyRoad
    det, ins, seg = _scene(seed=5)
    seg[..., 1] = 0                                          # theta = 0 -> width clipped to 1 -> unit = road size
    H, W = seg.shape[1:3]
    masks = O.crop_and_pad_mask((H, W), det, ins)
    want = O.calculate_instance_size(seg, masks, 3.25)
    got = host(CalculateInstanceSize(3.25)([dev(seg), dev(masks)]))
    np.testing.assert_allclose(got, want, rtol=1e-5)
    np.testing.assert_array_equal(host(IncludeMyRoad()([dev(seg), dev(masks)])), O.include_my_road(seg, masks))


def test_serving_model_end_to_end():
    """uint8 images -> deploy model -> CropAndPadMask -> SummaryOutput against the oracle's serving_forward."""
    from masklab_hip import ModelConfiguration, retinamasklab as R
    cfg = ModelConfiguration()
    cfg.backbone.backbone_type = "mobilenet"
    cfg.postprocess.resolution = (128, 256)
    _, model = R.construct_masklab_networks(cfg)
    w = model.init_weights(3)
    for k in w:
        if k.startswith("classification_sub_net/") and k.endswith("/output/kernel"):
            w[k] = (w[k] * 8.0).astype(np.float32)
    model.load_weights(w, "cuda:0")
    serving = R.construct_serving_network(cfg, R.construct_deploy_network(cfg, model))
    images = np.random.default_rng(1234).integers(0, 256, (2, 320, 640, 3), dtype=np.uint8)
    got = serving.predict(images)
    np.testing.assert_array_equal(got, host(serving(dev(images), materialise_masks=True)))   # the reference's literal wiring
    want = O.serving_forward(cfg, w, images, literal_groups=False)
    assert got.shape == want.shape and got.shape[-1] == 11
    np.testing.assert_array_equal(got[..., 0], want[..., 0])                     # classes / padding rows
    assert np.abs(got[..., 1:6] - want[..., 1:6]).max() <= 1                     # int-truncated boxes (see test_gpu_deploy)
    # areas follow the thresholded masks: a flipped mask pixel changes a count by one
    np.testing.assert_allclose(got[..., 6], want[..., 6], rtol=2e-3, atol=2)
    np.testing.assert_allclose(got[..., 7:10], want[..., 7:10], rtol=1e-2, atol=1e-2 * max(1.0, float(want[..., 7:10].max())))
