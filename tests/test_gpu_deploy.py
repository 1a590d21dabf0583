"""GPU parity of the deploy wrapper either side of the forward (reference
engine/retinamasklab.py:598-643): DownSampleInput, TrimInstances, SemanticSmoothing, UpSampleOutput
against the CPU oracle.  Integer / index outputs bit-exact on identical inputs; the end-to-end test
allows flips only where the oracle's float value sits within 1e-3 of a threshold.  -m gpu."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import masklab as O
from oracle import tfops as T


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def host(t):
    torch.cuda.synchronize()
    return t.cpu().numpy()


@pytest.mark.parametrize("shape,target", [((2, 97, 131, 3), (54, 96)), ((1, 1080, 1920, 3), (540, 960)),
                                          ((3, 64, 48, 3), (540, 960)), ((1, 33, 33, 1), (33, 33))])
@pytest.mark.parametrize("u8", [True, False])
def test_down_sample_input(shape, target, u8):
    from masklab_hip.layers import DownSampleInput
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    if not u8:
        img = (img.astype(np.float32) + rng.random(shape, dtype=np.float32)).astype(np.float32)
    want = O.down_sample_input(img, target)
    got = host(DownSampleInput(target)(dev(img)))
    assert got.shape == want.shape and got.dtype == np.float32
    np.testing.assert_array_equal(got, want)          # same fp32 expression order, contraction off


def _molded_rois(B, N, C, hw, seed, holes=False):
    rng = np.random.default_rng(seed)
    boxes = np.full((B, N, 6), -1, np.float32)
    for b in range(B):
        n = int(rng.integers(0, N + 1)) if b else N // 2
        rows = np.arange(N) < n
        if holes:                                       # TrimInstances itself makes no assumption on row order
            rows = rng.random(N) < 0.5
        k = int(rows.sum())
        boxes[b, rows, :4] = rng.uniform(5, 400, (k, 4)).astype(np.float32)
        boxes[b, rows, 4] = rng.integers(0, C, k).astype(np.float32)
        boxes[b, rows, 5] = rng.uniform(0.5, 1.0, k).astype(np.float32)
    masks = rng.random((B, N, hw, hw, C), dtype=np.float32)
    masks[boxes[:, :, 4] == -1] = -1
    return boxes, masks


@pytest.mark.parametrize("B,N,holes", [(3, 100, False), (2, 37, True), (1, 300, True), (2, 8, False)])
def test_trim_instances(B, N, holes):
    from masklab_hip.layers import TrimInstances
    boxes, masks = _molded_rois(B, N, 5, 28, seed=N, holes=holes)
    wb, wm = O.trim_instances(boxes, masks, mold=True)
    gb, gm = TrimInstances(mold=True)([dev(boxes), dev(masks)])
    np.testing.assert_array_equal(host(gb), wb)
    np.testing.assert_array_equal(host(gm), wm)
    fb, fm = O.trim_instances(boxes, masks, mold=False)
    gb, gm = TrimInstances(mold=False)([dev(boxes), dev(masks)])
    np.testing.assert_array_equal(host(gb), fb)
    np.testing.assert_array_equal(host(gm), fm)


def test_trim_instances_no_detection_at_all():
    from masklab_hip.layers import TrimInstances
    boxes = np.full((2, 10, 6), -1, np.float32)
    masks = np.full((2, 10, 28, 28, 5), -1, np.float32)
    wb, wm = O.trim_instances(boxes, masks)
    gb, gm = TrimInstances()([dev(boxes), dev(masks)])
    assert wb.shape == (2, 1, 6) and wm.shape == (2, 1, 28, 28)            # MoldBatch keeps one padded row
    np.testing.assert_array_equal(host(gb), wb)
    np.testing.assert_array_equal(host(gm), wm)


@pytest.mark.parametrize("k", [0, 1, 2, 3, 10, 11, 40])
def test_semantic_smoothing_layer(k):
    from masklab_hip.layers import SemanticSmoothing
    x = np.random.default_rng(k).random((2, 23, 31, 3), dtype=np.float32)
    want = O.semantic_smoothing(x, k, 1.5)
    got = host(SemanticSmoothing(kernel_size=k, weight=1.5)(dev(x)))
    np.testing.assert_array_equal(got, want)           # min / max / one multiply: exact


def test_semantic_smoothing_per_class():
    from masklab_hip.layers import SemanticSmoothing
    x = np.random.default_rng(0).random((1, 68, 120, 3), dtype=np.float32)
    ks, ws = (11, 0, 4), (1.0, 2.0, 0.5)
    want = np.concatenate([O.semantic_smoothing(x[..., i:i + 1], k, w) for i, (k, w) in enumerate(zip(ks, ws))], -1)
    got = host(SemanticSmoothing.smooth_classes(dev(x), ks, ws))
    np.testing.assert_array_equal(got, want)


def test_up_sample_output():
    from masklab_hip.layers import UpSampleOutput
    rng = np.random.default_rng(2)
    boxes, _ = _molded_rois(2, 20, 5, 4, seed=9)
    masks = rng.random((2, 20, 28, 28), dtype=np.float32)
    masks[boxes[:, :, 4] == -1] = -1
    sem = rng.random((2, 54, 96, 3), dtype=np.float32)
    target = np.zeros((2, 211, 377, 3), np.uint8)
    wb, wm, ws = O.up_sample_output(boxes, masks, sem, target.shape[1:3])
    gb, gm, gs = UpSampleOutput()([dev(boxes), dev(masks), dev(sem)], target=dev(target))
    for g, w_ in ((gb, wb), (gm, wm), (gs, ws)):
        g = host(g)
        assert g.dtype == np.int32 and g.shape == w_.shape
        np.testing.assert_array_equal(g, w_)


def test_resize_like_on_three_channel_map():
    from masklab_hip.layers import ResizeLike
    x = np.random.default_rng(1).random((2, 17, 29, 3), dtype=np.float32)
    tgt = np.zeros((2, 135, 230, 3), np.float32)
    got = host(ResizeLike()(dev(x), target=dev(tgt)))
    np.testing.assert_array_equal(got, T.resize_bilinear_align_corners(x, 135, 230))


@pytest.mark.parametrize("bt,resolution,ishape", [
    ("mobilenet", (128, 256), (2, 320, 640, 3)),        # MobileNet's explicit (0,1) pads need /128 working sizes
    ("resnext50", (100, 180), (2, 200, 360, 3)),        # 'same' everywhere: odd level sizes 25, 13, 7, ... 45, 23
])
def test_deploy_model_end_to_end(bt, resolution, ishape):
    """images of one resolution -> DownSampleInput to the working size -> full forward -> int32
    (detection, instance, semantic) at the input resolution, against the oracle's deploy_forward."""
    from masklab_hip import ModelConfiguration, retinamasklab as R
    cfg = ModelConfiguration()
    cfg.backbone.backbone_type = bt
    cfg.postprocess.resolution = resolution
    cfg.postprocess.smoothing_kernel_sizes = (5, 0, 3)
    cfg.postprocess.smoothing_weights = (1.0, 1.25, 0.75)
    _, model = R.construct_masklab_networks(cfg)
    w = model.init_weights(3)
    for k in w:
        if k.startswith("classification_sub_net/") and k.endswith("/output/kernel"):
            w[k] = (w[k] * 8.0).astype(np.float32)          # some anchors pass min_confidence
    model.load_weights(w, "cuda:0")
    deploy = R.construct_deploy_network(cfg, model)
    images = np.random.default_rng(1234).integers(0, 256, ishape, dtype=np.uint8)
    det, inst, sem = deploy.predict(images)
    wdet, winst, wsem = O.deploy_forward(cfg, w, images, literal_groups=False)
    assert det.dtype == inst.dtype == sem.dtype == np.int32
    assert det.shape == wdet.shape and inst.shape == winst.shape and sem.shape == wsem.shape == ishape
    assert (wdet[..., 4] >= 0).sum() > 0, "fixture produced no detections"
    assert 0 < wsem.mean() < 1 and 0 < winst.mean() < 1, "fixture thresholds are degenerate"
    np.testing.assert_array_equal(det[..., 4], wdet[..., 4])                   # labels and padding pattern
    assert np.abs(det - wdet).max() <= 1                                       # truncation of x*ratio at an integer
    assert (det != wdet).mean() < 0.02
    assert (inst != winst).mean() < 1e-3 and (sem != wsem).mean() < 1e-3      # flips only at |v - 0.5| < 1e-3


def test_deploy_model_needs_all_heads():
    from masklab_hip import ModelConfiguration, retinamasklab as R
    cfg = ModelConfiguration()
    cfg.backbone.backbone_type = "mobilenet"
    bb = R.build_backbone_network(cfg)
    model = R.construct_inference_network(cfg, bb, detection_networks=R.build_detection_network(cfg))
    with pytest.raises(ValueError):
        R.construct_deploy_network(cfg, model)
