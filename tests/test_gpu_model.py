"""End-to-end GPU parity: the MI355X InferenceModel against the CPU oracle forward, same synthetic
weights, same seeded images.  Tolerance from BASELINE.json north_star: float outputs within 1e-3
(fp32), box / class indices bit-exact.  -m gpu."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import masklab as O

TOL = 1e-3


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _build(bt, seed=3, hot_cls=False):
    from masklab_hip import ModelConfiguration, retinamasklab as R
    cfg = ModelConfiguration()
    cfg.backbone.backbone_type = bt
    _, model = R.construct_masklab_networks(cfg)
    w = model.init_weights(seed)
    if hot_cls:
        # random-init scores sit at ~0.01 (bias -log 99): widen the class logits so that a fraction
        # of a percent of (anchor,class) pairs pass min_confidence=0.5 and the NMS stages do real work
        for k in w:
            if k.startswith("classification_sub_net/") and k.endswith("/output/kernel"):
                w[k] = (w[k] * 8.0).astype(np.float32)
    model.load_weights(w, "cuda:0")
    return cfg, model, w


def _check(model, got, want):
    for name, g, r in zip(model.output_names, got, want):
        assert g.shape == r.shape, (name, g.shape, r.shape)
        if name == "roi_boxes":
            np.testing.assert_array_equal(g[..., 4], r[..., 4], err_msg="class ids")
            np.testing.assert_array_equal(g == -1, r == -1, err_msg="padding pattern")
            # pixel coordinates are O(100): fp32 relative tolerance; confidences absolute
            np.testing.assert_allclose(g[..., :4], r[..., :4], rtol=1e-5, atol=TOL)
            np.testing.assert_allclose(g[..., 5], r[..., 5], rtol=0, atol=TOL)
            continue
        err = float(np.max(np.abs(g.astype(np.float64) - r))) if g.size else 0.0
        assert err <= TOL, (name, err)


@pytest.mark.parametrize("bt", ["mobilenet", "resnext50", "resnext101"])
def test_full_forward_matches_oracle(bt):
    cfg, model, w = _build(bt)
    images = np.random.default_rng(1234).integers(0, 256, (2, 128, 128, 3), dtype=np.uint8)
    got = model.predict(images)
    want = O.inference_forward(cfg, w, images, literal_groups=False)
    _check(model, got, want)


@pytest.mark.parametrize("bt", ["mobilenet", "resnext50"])
def test_full_forward_with_detections(bt):
    cfg, model, w = _build(bt, seed=5, hot_cls=True)
    images = np.random.default_rng(99).integers(0, 256, (2, 128, 256, 3), dtype=np.uint8)
    # Thresholding is discontinuous: pick min_confidence in the widest score gap near 0.5 so that
    # fp32 reordering (|diff| ~1e-6) cannot move a score across it (SURVEY.md section 7 hard parts).
    cls_ref = O.inference_forward(cfg, w, images, literal_groups=False, with_instance=False,
                                  with_semantic=False)[0]
    s = np.sort(cls_ref[(cls_ref > 0.45) & (cls_ref < 0.55)].astype(np.float64))
    gaps = np.diff(s)
    i = int(np.argmax(gaps))
    thr = float(np.float32((s[i] + s[i + 1]) / 2))
    assert gaps[i] > 2e-4, "no usable gap in the score distribution"
    cfg.detection.min_confidence = thr
    model.detection_proposal.min_confidence = thr
    got = model.predict(images, want_kept=True)
    want, internals = O.inference_forward(cfg, w, images, literal_groups=False, return_internals=True)
    kept_ref = internals["kept"]
    assert len(kept_ref) > 0, "fixture produced no detections; raise the logit scale"
    det = model.last_detections
    counts = det["counts"].cpu().numpy()
    kept = det["kept"].cpu().numpy()
    for b in range(images.shape[0]):
        np.testing.assert_array_equal(kept[b, :counts[b]], kept_ref[kept_ref[:, 0] == b][:, 1:])
    _check(model, got, want)
    # the reference's own detection metric (engine/metrics.py:109-165), oracle detections as
    # ground truth -- the stand-in for BASELINE's "box AP vs Keras ref".  Same detections => 1.0.
    from oracle import metrics as OM
    names = model.output_names
    pr, rc, fm = OM.detection_iou_metric(got[names.index("roi_boxes")], want[names.index("roi_boxes")])
    np.testing.assert_allclose(pr, 1.0, atol=1e-6)
    np.testing.assert_allclose(rc, 1.0, atol=1e-6)
    np.testing.assert_allclose(fm, 1.0, atol=1e-6)


def test_heads_optional_like_reference():
    from masklab_hip import ModelConfiguration, retinamasklab as R
    cfg = ModelConfiguration()
    cfg.backbone.backbone_type = "mobilenet"
    R.K.clear_session()
    bb = R.build_backbone_network(cfg)
    sem = R.build_semantic_network(cfg)
    model = R.construct_inference_network(cfg, bb, semantic_networks=sem)
    assert model.output_names == ["seg_pred"]
    w = model.init_weights(0)
    model.load_weights(w, "cuda:0")
    images = np.random.default_rng(0).integers(0, 256, (1, 128, 128, 3), dtype=np.uint8)
    (seg,) = model.predict(images)
    (ref,) = O.inference_forward(cfg, w, images, with_detection=False)
    assert float(np.max(np.abs(seg - ref))) <= TOL


@pytest.mark.parametrize("case", ["forward_mobilenet_128", "forward_resnext50_128"])
def test_forward_matches_committed_golden(case, golden_dir):
    """HIP path vs the committed end-to-end vectors (tests/golden/*.npz): indices bit-exact,
    float outputs within 1e-3; the oracle is NOT run here."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("mk", os.path.join(golden_dir, "make_forward_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    g = np.load(os.path.join(golden_dir, case + ".npz"))
    cfg, model, w, images = mk.build_case(*mk.CASES[case])
    cfg.detection.min_confidence = float(g["min_confidence"])
    model.detection_proposal.min_confidence = float(g["min_confidence"])
    model.load_weights(w, "cuda:0")
    got = model.predict(images, want_kept=True)
    det = model.last_detections
    counts, kept = det["counts"].cpu().numpy(), det["kept"].cpu().numpy()
    kept_ref = g["kept"]
    for b in range(images.shape[0]):
        np.testing.assert_array_equal(kept[b, :counts[b]], kept_ref[kept_ref[:, 0] == b][:, 1:])
    _check(model, got, [g[n] for n in model.output_names])


def test_batch_sharding_equals_full_batch():
    """data parallel = split the batch, replicate weights, concatenate (reference parallel.py:64-107):
    per-image results do not depend on which shard an image is in."""
    from masklab_hip import parallel
    cfg, model, w = _build("mobilenet", seed=7, hot_cls=True)
    images = torch.from_numpy(np.random.default_rng(5).integers(0, 256, (4, 128, 128, 3), dtype=np.uint8))
    model.call(images.cuda())
    full = {k: v.clone() for k, v in model.last_detections.items() if v is not None}
    seg_full = model.call(images.cuda())[-1].clone()
    parts, segs = [], []
    for r in range(2):
        outs = model.call(parallel.shard_batch(images, r, 2).cuda())
        parts.append({k: v.clone() for k, v in model.last_detections.items() if v is not None})
        segs.append(outs[-1].clone())
    torch.cuda.synchronize()
    assert torch.equal(torch.cat([p["counts"] for p in parts]), full["counts"])
    assert torch.equal(torch.cat([p["proposed"] for p in parts]), full["proposed"])
    assert torch.equal(torch.cat(segs), seg_full)


def test_full_forward_with_squeeze_excite_and_separable_conv():
    """the optional tower variants (flags off by default; the reference's shipped
    project config turns SqueezeExcite on).  Builder quirks kept: the box tower's SE flag is
    `use_separable_conv`, mask/seg heads get expand_ratio = use_separable_conv (= 1)."""
    from masklab_hip import ModelConfiguration, retinamasklab as R
    cfg = ModelConfiguration()
    cfg.backbone.backbone_type = "mobilenet"
    cfg.detection.use_separable_conv = True
    cfg.detection.use_squeeze_excite = True
    cfg.instance.use_separable_conv = True
    cfg.instance.use_squeeze_excite = True
    cfg.semantic.use_squeeze_excite = True        # separable decoder cannot build: 160-ch input vs 128 filters
    cfg.detection.min_confidence = 0.02
    _, model = R.construct_masklab_networks(cfg)
    w = model.init_weights(4)
    model.load_weights(w, "cuda:0")
    images = np.random.default_rng(77).integers(0, 256, (2, 128, 128, 3), dtype=np.uint8)
    got = model.predict(images)
    want = O.inference_forward(cfg, w, images)
    _check(model, got, want)
    cfg.semantic.use_separable_conv = True
    with pytest.raises(ValueError):
        R.construct_masklab_networks(cfg)


def _shipped_head_config(bt):
    """The head configuration the reference project ships (road_project/train.py:36-58) on a backbone this build supports
    (the project's 'seresnet34' needs the un-vendored keras_applications): FOUR pyramid levels (C3, C4, C5, P6 -- no
    P7), tower depth 3, prior ratios 1/2, 1, 2, 5, 8, SqueezeExcite in every head."""
    from masklab_hip import ModelConfiguration
    cfg = ModelConfiguration()
    cfg.backbone.backbone_type = bt
    cfg.backbone.backbone_outputs = ('C3', 'C4', 'C5', 'P6')
    cfg.detection.num_features = 128
    cfg.detection.num_depth = 3
    cfg.detection.use_squeeze_excite = True
    cfg.detection.pr_scales = [2 ** 0, 2 ** (1 / 3), 2 ** (2 / 3)]
    cfg.detection.pr_ratios = [1 / 2, 1, 2, 5, 8]
    cfg.instance.crop_size = (14, 14)
    cfg.instance.max_k = 2
    cfg.instance.num_features = 128
    cfg.instance.num_depth = 4
    cfg.instance.use_squeeze_excite = True
    cfg.semantic.num_features = 128
    cfg.semantic.num_depth = 3
    cfg.semantic.use_squeeze_excite = True
    return cfg


@pytest.mark.parametrize("bt,shape", [("resnext50", (2, 256, 384, 3)), ("mobilenet", (3, 128, 256, 3))])
def test_full_forward_on_the_reference_projects_shipped_head_config(bt, shape):
    """The configuration the reference actually trains and serves (road_project/train.py:36-58), end to end against the
    oracle with detections: the multi-problem tower launches, the GroupNorm multi launches and the level molding see
    FOUR pyramid levels (4 tower problems per launch, no P7) and depth-3 towers with SqueezeExcite here -- every other
    end-to-end test runs the 5-level / depth-4 defaults.  Indices bit-exact, floats within 1e-3; also through the
    fixed-capacity stage 2 (no host read) and the whole-forward hipGraph."""
    from masklab_hip import retinamasklab as R
    cfg = _shipped_head_config(bt)
    _, model = R.construct_masklab_networks(cfg)
    assert model.backbone_network.output_names == ['C3', 'C4', 'C5', 'P6']
    from oracle import fixtures as FX
    w = model.init_weights(5)
    images = np.random.default_rng(shape[1] + shape[2]).integers(0, 256, shape, dtype=np.uint8)
    # an order-stable detection fixture (oracle/fixtures.py): class logits scaled so that scores pass 0.5 WITHOUT saturating,
    # min_confidence in a score gap, kept list unchanged under 3e-5 score noise (x8 logits pile scores up near 1.0 and rows
    # of near-equal score may legitimately swap)
    c1, l1 = O.inference_forward(cfg, w, images, literal_groups=False, with_instance=False, with_semantic=False)
    assert c1.shape[1] == 15 * sum(-(-shape[1] // s) * -(-shape[2] // s) for s in (8, 16, 32, 64))   # four levels of anchors
    scale, thr = FX.choose_logit_scale(cfg, c1, l1, shape[1], shape[2])
    assert scale is not None, "no order-stable logit scale on the grid"
    w = FX.scale_cls_logits(w, scale)
    model.load_weights(w, "cuda:0")
    cfg.detection.min_confidence = thr
    model.detection_proposal.min_confidence = thr
    want, internals = O.inference_forward(cfg, w, images, literal_groups=False, return_internals=True)
    kept_ref = internals["kept"]
    assert len(kept_ref) > 0, "fixture produced no detections"
    got = model.predict(images, want_kept=True)
    det = model.last_detections
    counts, kept = det["counts"].cpu().numpy(), det["kept"].cpu().numpy()
    for b in range(shape[0]):
        np.testing.assert_array_equal(kept[b, :counts[b]], kept_ref[kept_ref[:, 0] == b][:, 1:])
    _check(model, got, want)
    model.device_counts = True                       # stage 2 at capacity, no host read inside the forward
    _check(model, model.predict(images), want)
    model.device_counts = "auto"
    model.enable_graphs(True)                        # ... and as ONE hipGraph
    for _ in range(2):
        _check(model, model.predict(images), want)
    model.enable_graphs(False)


@pytest.mark.parametrize("bt", ["mobilenet", "resnext50"])
def test_hipgraph_replay_equals_eager(bt):
    """enable_graphs(): stage 1 captured into a hipGraph and replayed gives bit-identical outputs to the eager
    launches (same kernels, same order), for the image it was captured on and for a different one."""
    cfg, model, w = _build(bt, seed=5, hot_cls=True)
    rng = np.random.default_rng(7)
    imgs = [rng.integers(0, 256, (1, 128, 256, 3), dtype=np.uint8) for _ in range(3)]
    host_read = [model.predict(im) for im in imgs]         # stage 2 sized by the host's read of the RoI counts
    model.device_counts = True                             # the same launches the graph captures, issued eagerly
    eager = [model.predict(im) for im in imgs]
    model.device_counts = "auto"
    for a, b in zip(host_read, eager):
        for name, x, y in zip(model.output_names, a, b):
            assert x.shape == y.shape, name                # molded exactly like MoldBatch (reference misc.py:231-286)
            if name == "roi_masks":                        # (a launch at capacity may cut its K sum elsewhere: fp32 rounding)
                np.testing.assert_allclose(x, y, atol=1e-5, err_msg=name)
            else:
                np.testing.assert_array_equal(x, y, err_msg=name)
    model.enable_graphs(True)
    for rep in range(2):                                   # first pass captures, second only replays
        for im, want in zip(imgs, eager):
            got = model.predict(im)
            for name, g, r in zip(model.output_names, got, want):
                np.testing.assert_array_equal(g, r, err_msg=f"{name} (pass {rep})")
    assert len(model._graphs) == 1
    assert next(iter(model._graphs))[3] is True        # the WHOLE forward is the graph (fixed-capacity stage 2, no host read)
    other = rng.integers(0, 256, (2, 128, 128, 3), dtype=np.uint8)     # a second shape gets its own graph
    model.enable_graphs(False)
    want = model.predict(other)
    model.enable_graphs(True)
    got = model.predict(other)
    for name, g, r in zip(model.output_names, got, want):
        if name == "roi_masks":                            # (eager = RoI batches sized by the host read, graph = at capacity: see above)
            np.testing.assert_allclose(g, r, atol=1e-5, err_msg=name)
        else:
            np.testing.assert_array_equal(g, r, err_msg=name)
    model.enable_graphs(False)


@pytest.mark.parametrize("bt,shape", [
    ("resnext50", (3, 192, 320, 3)),        # non-square, level sizes 24x40 .. 2x3 (odd at P6/P7)
    ("resnext50", (1, 136, 200, 3)),        # not a multiple of the strides: 'same' ceil sizes 17x25, 9x13, 5x7, 3x4, 2x2
    ("resnext50", (2, 296, 280, 3)),        # stage-2 maps of 74x70 >= 4096 pixels: the pipelined 1x1 kernel on a ragged M
                                            # (10 360 rows = 80.9 panels), odd level sizes 37x35, 19x18, 10x9, 5x5, 3x3
    ("mobilenet", (5, 128, 384, 3)),        # odd batch
    ("mobilenet", (32, 128, 128, 3)),       # the MoldBatch maximum (reference misc.py:275)
])
def test_full_forward_other_shapes(bt, shape):
    cfg, model, w = _build(bt, seed=11, hot_cls=True)
    images = np.random.default_rng(shape[0] * 7 + shape[2]).integers(0, 256, shape, dtype=np.uint8)
    # thresholding is discontinuous: put min_confidence in the widest score gap near 0.6 (see above)
    cls_ref = O.inference_forward(cfg, w, images, literal_groups=False, with_instance=False, with_semantic=False)[0]
    sc = np.sort(cls_ref[(cls_ref > 0.55) & (cls_ref < 0.65)].astype(np.float64))
    gaps = np.diff(sc)
    i = int(np.argmax(gaps))
    assert gaps[i] > 2e-5, "no usable gap in the score distribution"      # GPU-vs-oracle score differences are ~1e-6
    thr = float(np.float32((sc[i] + sc[i + 1]) / 2))
    cfg.detection.min_confidence = thr
    model.detection_proposal.min_confidence = thr
    got = model.predict(images)
    want = O.inference_forward(cfg, w, images, literal_groups=False)
    assert (want[2][..., 4] >= 0).sum() > 0, "fixture produced no detections"
    _check(model, got, want)


def test_more_than_32_images_is_refused_like_the_reference():
    cfg, model, w = _build("mobilenet", seed=1)
    images = np.zeros((33, 128, 128, 3), np.uint8)
    with pytest.raises(ValueError, match="32"):
        model.predict(images)


@pytest.mark.parametrize("bt", ["mobilenet", "resnext50"])
def test_side_stream_semantic_head_is_bit_identical(bt):
    """use_side_stream: the semantic head runs on a second HIP stream beside FPN / towers / detection (own scratch
    buffers per stream, joined before stage 1 returns) -- same kernels, same inputs, so the same bits; also under
    hipGraph capture (the side stream forks from and joins the capturing stream)."""
    cfg, model, w = _build(bt, seed=5, hot_cls=True)
    rng = np.random.default_rng(17)
    imgs = [rng.integers(0, 256, (2, 128, 256, 3), dtype=np.uint8) for _ in range(3)]
    model.use_side_stream = False
    want = [model.predict(im) for im in imgs]
    model.use_side_stream = True
    for rep in range(3):
        for im, ref in zip(imgs, want):
            got = model.predict(im)
            for name, g, r in zip(model.output_names, got, ref):
                np.testing.assert_array_equal(g, r, err_msg=f"{name} (side stream, pass {rep})")
    model.enable_graphs(True)
    for rep in range(2):
        for im, ref in zip(imgs, want):
            got = model.predict(im)
            for name, g, r in zip(model.output_names, got, ref):
                np.testing.assert_array_equal(g, r, err_msg=f"{name} (side stream + graph, pass {rep})")
    model.enable_graphs(False)


_HEADLINE = {}


def _headline_fixture(bt, size):
    """Oracle side of the headline-size fixture, computed once per (backbone, size) and shared by the tests at that size
    (one oracle forward at 1024^2 / 1280^2 costs 10-40 s): seed-0 weights with the class logits scaled by
    FX.KNOWN_SCALE, the default_rng(1234) image, min_confidence in a score gap, order stability under 3e-5 noise."""
    from oracle import fixtures as FX
    if (bt, size) not in _HEADLINE:
        from masklab_hip import ModelConfiguration, retinamasklab as R
        cfg = ModelConfiguration()
        cfg.backbone.backbone_type = bt
        _, model = R.construct_masklab_networks(cfg)
        w_fix = FX.scale_cls_logits(model.init_weights(0), FX.KNOWN_SCALE[(bt, size)])
        image = np.random.default_rng(1234).integers(0, 256, (1, size, size, 3), dtype=np.uint8)
        want, internals = O.inference_forward(cfg, w_fix, image, literal_groups=False, return_internals=True,
                                              min_confidence=lambda c: FX.gap_threshold(c)[0])
        thr = internals["min_confidence"]
        names = model.output_names
        cls_ref, loc_ref = want[names.index("cls_pred")], want[names.index("loc_pred")]
        assert FX.gap_threshold(cls_ref)[1] > 2e-4, "min_confidence does not sit in a usable score gap"
        assert float(cls_ref.max()) < 0.93, "scores saturate: the fixture would contain near-ties"
        kept_ref, stable = FX.order_stability(cfg, cls_ref, FX.boxes_from(cfg, loc_ref, size, size), thr, trials=8)
        assert stable == 8 and len(kept_ref) >= 30, (stable, len(kept_ref))
        np.testing.assert_array_equal(kept_ref, internals["kept"])
        _HEADLINE[(bt, size)] = dict(w_fix=w_fix, image=image, want=want, thr=thr, kept=kept_ref)
    return _HEADLINE[(bt, size)]


@pytest.mark.parametrize("bt,size", [("resnext50", 1024), ("resnext101", 1280)])
def test_headline_size_indices_bit_exact(bt, size):
    """north_star: "bit-exact box/class indices" AT THE HEADLINE SIZE (BASELINE configs[2]: ResNeXt-50 1024^2;
    configs[4] shape: ResNeXt-101 1280^2, fp32 path).  One image, seed-0 weights, the class logits scaled so that a
    few hundred (anchor, class) scores pass 0.5 WITHOUT saturating (oracle/fixtures.py), min_confidence in a score gap.
    Asserts: the fixture is order-stable under 3x the GPU deviation, the kept (anchor, class) list equals the oracle's
    IN ORDER (reference engine/layers/detection.py:491-563), and every output is within tolerance."""
    fx = _headline_fixture(bt, size)
    cfg, model, w = _build(bt, seed=0)
    model.reload_class_outputs(fx["w_fix"])
    cfg.detection.min_confidence = fx["thr"]
    model.detection_proposal.min_confidence = fx["thr"]
    got = model.predict(fx["image"], want_kept=True)
    det = model.last_detections
    n = int(det["counts"].cpu()[0])
    np.testing.assert_array_equal(det["kept"].cpu().numpy()[0, :n], fx["kept"][:, 1:])      # same rows, same ORDER
    _check(model, got, fx["want"])


def _launch_tiles(label):
    """128 x 128 tiles of a logged conv launch (ops.LAUNCH_LOG label: "M=.. N=.." per problem)."""
    import re
    return sum(-(-int(m) // 128) * -(-int(n) // 128) for m, n in re.findall(r"M=(\d+) N=(\d+)", label))


def test_full_per_gpu_batch_of_the_headline_config():
    """BASELINE configs[2] / [3]: the WHOLE per-GPU batch -- ResNeXt-50, 8 x 1024 x 1024, fp32 -- through the whole model in
    ONE forward (what bench.py times), not one image at a time:
      * image 0 of the BATCH against the oracle on the order-stable fixture: kept (anchor, class) rows and their ORDER
        exact, floats within 1e-3 (reference engine/retinamasklab.py:420-495);
      * image k of the batch against the same image run ALONE.  A 1-image launch of few tiles is cut along K where the
        8-image launch is not (csrc/conv_mfma.hip choose_splits: < 192 tiles), so the two sum K in other pieces: the
        launches that do are NAMED (ops.LAUNCH_LOG, ml_conv2d_launch_splits) and every one of them must be such a small
        launch; results then agree to fp32 rounding, the detections row for row."""
    from masklab_hip import ops
    from oracle import fixtures as FX
    bt, size, B = "resnext50", 1024, 8
    fx = _headline_fixture(bt, size)
    cfg, model, w = _build(bt, seed=0)
    model.reload_class_outputs(fx["w_fix"])
    cfg.detection.min_confidence = fx["thr"]
    model.detection_proposal.min_confidence = fx["thr"]
    images = np.random.default_rng(1234).integers(0, 256, (B, size, size, 3), dtype=np.uint8)
    assert np.array_equal(images[:1], fx["image"])            # image 0 of the batch IS the fixture image
    names = model.output_names
    ops.LAUNCH_LOG = []
    try:
        outs = model.predict(images, want_kept=True)
        log_batch, ops.LAUNCH_LOG = ops.LAUNCH_LOG, []
        det = model.last_detections
        lcounts = det["level_counts"].cpu().numpy()
        counts, kept = det["counts"].cpu().numpy(), det["kept"].cpu().numpy()
        # ---- image 0 of the batch vs the oracle
        one = FX.image_of_batch(names, outs, lcounts, 0)
        np.testing.assert_array_equal(kept[0, :counts[0]], fx["kept"][:, 1:])
        _check(model, one, fx["want"])
        # ---- image k of the batch vs image k alone
        for k in (0, 5):
            ops.LAUNCH_LOG = []
            alone = model.predict(images[k:k + 1], want_kept=True)
            log_one = ops.LAUNCH_LOG
            d1 = model.last_detections
            assert len(log_one) == len(log_batch)
            differ = [(lo, sb, so) for (lb, sb), (lo, so) in zip(log_batch, log_one) if sb != so]
            assert differ, "expected the 1-image launches of the deep stages to be cut along K"
            for label, sb, so in differ:      # only launches the library's rule calls small for ONE image (< 192 tiles)
                assert _launch_tiles(label) < 192 and max(so) > 1, (label, sb, so)
            same = {lb for (lb, sb), (lo, so) in zip(log_batch, log_one) if sb == so}
            # stage 2 (the stem is the fused kernel since round 4: no conv2d launch, no K cut): the same sums either way
            assert any("HxW=256x256" in lb for lb in same) and not any("k7x7" in lb for lb, _ in log_batch)
            mine = FX.image_of_batch(names, outs, lcounts, k)
            n1 = int(d1["counts"].cpu()[0])
            floats = {"cls_pred", "loc_pred", "seg_pred"}
            for name, a, b in zip(names, mine, alone):
                if name in floats:
                    assert a.shape == b.shape and float(np.abs(a - b).max()) <= 5e-5, name
            if k == 0:
                # the fixture image (threshold in a score gap, no near-ties among the kept scores): the same rows in the same
                # order whichever launch shapes computed the scores, and the same RoI outputs
                np.testing.assert_array_equal(kept[k, :counts[k]], d1["kept"].cpu().numpy()[0, :n1])
                for name, a, b in zip(names, mine, alone):
                    assert a.shape == b.shape, name
                    if name == "roi_boxes":
                        np.testing.assert_array_equal(a[..., 4], b[..., 4])
                        np.testing.assert_allclose(a[..., :4], b[..., :4], rtol=1e-5, atol=1e-4)
                        np.testing.assert_allclose(a[..., 5], b[..., 5], rtol=0, atol=5e-5)
                    elif name == "roi_masks":
                        assert float(np.abs(a - b).max()) <= 5e-5, name
            else:
                # any other image: min_confidence was not placed in ITS score gaps, so a score within rounding of the
                # threshold (or of another candidate's) may legitimately come out differently in the two runs.  What must
                # hold for both: the detection stage is exact on the predictions it was given -- the oracle's
                # DetectionProposal (reference engine/layers/detection.py:482-567) on the GPU's own cls_pred / loc_pred
                # returns the GPU's rows, in order.
                det_cfg = cfg.detection
                for preds, kk, nn in ((mine, kept[k], int(counts[k])), (alone, d1["kept"].cpu().numpy()[0], n1)):
                    cp, lp = preds[names.index("cls_pred")], preds[names.index("loc_pred")]
                    _, kept_self = O.detection_proposal(cp, FX.boxes_from(cfg, lp, size, size), fx["thr"],
                                                        det_cfg.nms_iou_threshold, det_cfg.post_iou_threshold,
                                                        det_cfg.nms_max_output_size)
                    np.testing.assert_array_equal(kk[:nn], kept_self[:, 1:])
    finally:
        ops.LAUNCH_LOG = None


def test_batch_32_at_1024_crosses_2gib_activations():
    """The reference's MoldBatch takes up to 32 images per forward (engine/layers/misc.py:273-284).  At 1024x1024 the
    ResNeXt-50 stem output and every 256-channel stage-2 tensor is 2.1 GB: past the 32-bit buffer offsets of the
    generic conv kernel, which cuts such a problem into image groups (csrc/conv_mfma.hip split_by_image_groups); the
    persistent 1x1 kernel advances 64-bit bases per tile.  Stem -> max-pool -> stage 2 (-> P6 on the 2.1 GB C2 ->
    P7) for 32 images must equal the same images run one at a time bit for bit, and the oracle for the LAST image
    (the one that sits behind the 2 GiB mark)."""
    from masklab_hip import backbone as BB
    from masklab_hip import keras_like as K
    K.clear_session()
    bb = BB.load_backbone("resnext50", backbone_outputs=("C2", "P6", "P7"), num_features=128)
    w = K.init_weights(bb.weight_specs(), 0)
    bb.load_weights(w, torch.device("cuda:0"))
    B, S = 32, 1024
    images = np.random.default_rng(4).integers(0, 256, (B, S, S, 3), dtype=np.uint8)
    dev_images = torch.from_numpy(images).cuda()
    full = [t for t in bb(dev_images)]
    torch.cuda.synchronize()
    assert full[0].shape == (B, 256, 256, 256) and full[0].numel() * 4 >= 2 ** 31
    for b in (0, 17, 31):
        one = bb(dev_images[b:b + 1].contiguous())
        torch.cuda.synchronize()
        for name, f, o in zip(bb.output_names, full, one):
            if name == "C2":        # stem, max-pool, grouped 3x3 and the persistent 1x1 kernel: the same sums in any batch
                assert torch.equal(f[b:b + 1], o), f"{name}: image {b} differs between the 32-image and the 1-image run"
            else:                   # P6 / P7: a 1-image launch has few tiles and is split along K (other summation order)
                assert float((f[b:b + 1] - o).abs().max()) <= 2e-5, name
    names, ref = O.backbone_forward(images[31:32].astype(np.float32), w, "resnext50", ("C2", "P6", "P7"), literal_groups=False)
    assert names == bb.output_names
    for name, f, r in zip(names, full, ref):
        err = float(np.max(np.abs(f[31:32].cpu().numpy().astype(np.float64) - r)))
        assert err <= TOL, (name, err)


@pytest.mark.parametrize("size", [512, 1024])
def test_mobilenet_full_forward_at_config_size(size):
    """BASELINE configs[0] (MobileNet, 1x512x512) and configs[1] (MobileNet + FPN + ASPP, 1x1024x1024) AT THEIR SIZE:
    the full forward against the oracle on an order-stable fixture (class logits scaled so that scores pass 0.5 without
    saturating, min_confidence in a score gap, oracle/fixtures.py): float outputs within 1e-3, kept (anchor, class)
    rows and their ORDER exact (reference engine/backbone/base.py:253-258, engine/retinamasklab.py:420-495)."""
    from oracle import fixtures as FX
    cfg, model, w = _build("mobilenet", seed=0)
    images = np.random.default_rng(1234).integers(0, 256, (1, size, size, 3), dtype=np.uint8)
    c1, l1 = O.inference_forward(cfg, w, images, literal_groups=False, with_instance=False, with_semantic=False)
    scale, _ = FX.choose_logit_scale(cfg, c1, l1, size, size)
    assert scale is not None, "no order-stable logit scale on the grid"
    w_fix = FX.scale_cls_logits(w, scale)
    model.reload_class_outputs(w_fix)
    want, internals = O.inference_forward(cfg, w_fix, images, literal_groups=False, return_internals=True,
                                          min_confidence=lambda c: FX.gap_threshold(c)[0])
    thr = internals["min_confidence"]
    names = model.output_names
    cls_ref, loc_ref = want[names.index("cls_pred")], want[names.index("loc_pred")]
    kept_ref, stable = FX.order_stability(cfg, cls_ref, FX.boxes_from(cfg, loc_ref, size, size), thr, trials=8)
    assert stable == 8 and len(kept_ref) >= 4, (stable, len(kept_ref))
    cfg.detection.min_confidence = thr
    model.detection_proposal.min_confidence = thr
    got = model.predict(images, want_kept=True)
    det = model.last_detections
    n = int(det["counts"].cpu()[0])
    np.testing.assert_array_equal(det["kept"].cpu().numpy()[0, :n], kept_ref[:, 1:])
    _check(model, got, want)


def test_batch_sharding_where_the_split_k_decision_differs():
    """A shard and the full batch may take different split-K decisions (csrc/conv_mfma.hip choose_splits looks at the
    launch's tile count: one 256x256 image gives the five tower levels 11 tiles -> K cut into slices; eight images give
    88 tiles -> fewer slices).  The K sum is then cut at other places: results agree to fp32 rounding (2e-5), not bit
    for bit, and the detections are the same rows IN THE SAME ORDER -- on an order-stable fixture (oracle/fixtures.py:
    un-saturated scores, min_confidence in a score gap, kept list unchanged under 3e-5 score noise; with x8 logits
    dozens of scores saturate to within 1e-6 of each other and rows may legitimately swap)."""
    from masklab_hip import parallel
    from oracle import fixtures as FX
    cfg, model, w = _build("mobilenet", seed=7)
    images_np = np.random.default_rng(5).integers(0, 256, (8, 256, 256, 3), dtype=np.uint8)
    c1, l1 = O.inference_forward(cfg, w, images_np, literal_groups=False, with_instance=False, with_semantic=False)
    scale, thr = FX.choose_logit_scale(cfg, c1, l1, 256, 256)
    assert scale is not None, "no order-stable logit scale on the grid"
    model.reload_class_outputs(FX.scale_cls_logits(w, scale))
    cfg.detection.min_confidence = thr
    model.detection_proposal.min_confidence = thr
    images = torch.from_numpy(images_np)
    outs = [o.clone() for o in model.call(images.cuda(), want_kept=True)]
    full = {k: v.clone() for k, v in model.last_detections.items() if v is not None}
    assert int(full["counts"].sum()) >= 8, "fixture produced too few detections"
    parts, shard_outs = [], []
    for r in range(8):
        o = model.call(parallel.shard_batch(images, r, 8).cuda(), want_kept=True)
        parts.append({k: v.clone() for k, v in model.last_detections.items() if v is not None})
        shard_outs.append([t.clone() for t in o])
    torch.cuda.synchronize()
    names = model.output_names
    for n in ("cls_pred", "loc_pred", "seg_pred"):
        i = names.index(n)
        merged = torch.cat([so[i] for so in shard_outs])
        assert float((merged - outs[i]).abs().max()) <= 2e-5, n
    counts = torch.cat([p["counts"] for p in parts])
    assert torch.equal(counts, full["counts"])
    prop = torch.cat([p["proposed"] for p in parts])
    assert torch.equal(prop[..., 4], full["proposed"][..., 4])                   # same classes, same ORDER
    kept_parts = torch.cat([p["kept"] for p in parts])
    for b in range(8):                                                           # the same (anchor, class) rows, in order
        n = int(counts[b])
        assert torch.equal(kept_parts[b, :n], full["kept"][b, :n]), b
    assert float((prop - full["proposed"]).abs().max()) <= 1e-3                  # box pixels / scores to rounding


@pytest.mark.parametrize("bt,shape,thr", [("mobilenet", (3, 128, 256, 3), 0.5), ("resnext50", (2, 192, 160, 3), 0.5),
                                          ("mobilenet", (2, 128, 128, 3), 0.999)])
def test_fixed_capacity_stage2_matches_oracle(bt, shape, thr):
    """Stage 2 WITHOUT a host read (device_counts): RoI crops, mask-head convs / GroupNorms / fused tail launched for
    all nms_max_output_size slots of every level, the kernels skipping slots past the per-level maxima they read on the
    device; outputs molded afterwards.  Same result as the oracle (shapes = MoldBatch's dynamic N, indices exact),
    including images without detections (thr 0.999: every level molds to 1 padded row) and deferred outputs."""
    cfg, model, w = _build(bt, seed=5, hot_cls=True)
    images = np.random.default_rng(shape[1] + shape[2]).integers(0, 256, shape, dtype=np.uint8)
    if thr < 0.9:      # threshold in a score gap (see test_full_forward_with_detections)
        cls_ref = O.inference_forward(cfg, w, images, literal_groups=False, with_instance=False, with_semantic=False)[0]
        s = np.sort(cls_ref[(cls_ref > 0.45) & (cls_ref < 0.55)].astype(np.float64))
        i = int(np.argmax(np.diff(s)))
        thr = float(np.float32((s[i] + s[i + 1]) / 2))
    cfg.detection.min_confidence = thr
    model.detection_proposal.min_confidence = thr
    model.device_counts = True
    got = model.predict(images)
    want = O.inference_forward(cfg, w, images, literal_groups=False)
    _check(model, got, want)
    deferred = model(torch.from_numpy(images).cuda(), defer=True)
    assert type(deferred).__name__ == "DeferredOutputs"
    for g, r in zip(deferred.materialize(), got):
        np.testing.assert_array_equal(g.cpu().numpy(), r)


@pytest.mark.parametrize("math", ["f32", "f32x3"])
def test_fixed_capacity_tiles_that_cross_an_image_boundary_of_small_rois(math):
    """`live` launches skip tiles all of whose RoI slots are dead (csrc/conv_mfma.hip, deconv_out.hip).  With small crops a
    128-row tile spans THREE or more RoIs and can start in a dead slot of one image and end in a dead slot of the NEXT --
    with the live slot 0 of that image in between (7x7 crops = 49 rows, capacity 3: rows 128..255 hold slots 2 | 0 1 2 |
    0 1): the tile must run.  (Round 3's test of one period -- first slot <= last slot -- skipped it and left those masks
    unwritten.)  Against the oracle, fp32 and split-operand products."""
    from masklab_hip import ModelConfiguration, ops, retinamasklab as R
    cfg = ModelConfiguration()
    cfg.backbone.backbone_type = "mobilenet"
    cfg.instance.crop_size = (7, 7)
    cfg.detection.nms_max_output_size = 3
    _, model = R.construct_masklab_networks(cfg)
    w = model.init_weights(5)
    for k in w:
        if k.startswith("classification_sub_net/") and k.endswith("/output/kernel"):
            w[k] = (w[k] * 8.0).astype(np.float32)
    model.load_weights(w, "cuda:0")
    images = np.random.default_rng(21).integers(0, 256, (6, 128, 128, 3), dtype=np.uint8)
    cls_ref = O.inference_forward(cfg, w, images, literal_groups=False, with_instance=False, with_semantic=False)[0]
    sc = np.sort(cls_ref[(cls_ref > 0.45) & (cls_ref < 0.55)].astype(np.float64))
    i = int(np.argmax(np.diff(sc)))
    thr = float(np.float32((sc[i] + sc[i + 1]) / 2))
    cfg.detection.min_confidence = thr
    model.detection_proposal.min_confidence = thr
    want = O.inference_forward(cfg, w, images, literal_groups=False)
    n_img = (want[2][..., 4] >= 0).any(axis=1).sum()
    assert n_img >= 4, "the fixture needs detections in several consecutive images"
    model.device_counts = True
    assert model._capacity_wanted(torch.from_numpy(images))
    ops.set_conv_math(math)
    try:
        got = model.predict(images)
    finally:
        ops.set_conv_math("f32")
    _check(model, got, want)
