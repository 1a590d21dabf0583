"""CPU tests of the host side: the C ABI loads and exports every declared symbol, the drop-in
API surface (names, kwargs, get_config, error behaviour) mirrors the reference, weight naming /
packing is consistent with the oracle, golden fixtures are reproducible."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shared_library_exports_every_header_symbol():
    from masklab_hip import _lib
    header = open(os.path.join(ROOT, "include", "masklab_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(ml_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 20
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in masklab_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    assert lib.ml_version() == _lib.ABI_VERSION == 5
    assert lib.ml_conv2d_workspace_bytes() > 0 and lib.ml_groupnorm_workspace_bytes(8, 16) > 0
    assert lib.ml_detection_workspace_bytes(8, 327360, 5, 100) > 8 * 5 * 327360 * 24


def test_conv_desc_struct_matches_header_layout():
    import ctypes
    from masklab_hip import _lib
    # 5 pointers, 28 int32, 1 int64, then (ABI 5) the `live` pointer + 2 int32
    assert ctypes.sizeof(_lib.ConvDesc) == 5 * 8 + 28 * 4 + 8 + 8 + 2 * 4 + 8
    assert _lib.ConvDesc.out_bstride.offset == 152 and _lib.ConvDesc.math.offset == 144
    assert _lib.ConvDesc.live.offset == 160 and _lib.ConvDesc.live_period.offset == 168 and _lib.ConvDesc.gn_partials.offset == 176
    assert ctypes.sizeof(_lib.GnDesc) == 4 * 8 + 8 + 8 * 4 + 8 + 2 * 4 + 8 + 2 * 4 and _lib.GnDesc.live.offset == 72
    assert _lib.GnDesc.partials.offset == 88 and _lib.GnDesc.n_partials.offset == 96
    assert ctypes.sizeof(_lib.DeconvOutProblem) == 6 * 8 + 8 + 4 * 4 + 2 * 8 + 8 and _lib.DeconvOutProblem.live.offset == 88


def test_product_fails_loudly_without_gpu_tensors():
    torch = pytest.importorskip("torch")
    from masklab_hip import ops, packing
    dc_packed = packing.pack_dense(np.zeros((1, 1, 32, 8), np.float32))
    with pytest.raises((RuntimeError, AssertionError)):
        ops.conv2d(torch.zeros(1, 4, 4, 32), ops.DeviceConv(dc_packed, "cpu"))     # no CPU fallback


def test_model_configuration_matches_reference_defaults_and_roundtrips():
    from masklab_hip import ModelConfiguration
    c = ModelConfiguration()
    assert dir(c) == sorted(["postprocess", "backbone", "detection", "instance", "semantic", "loss", "dataset", "train"])
    assert c.backbone.backbone_outputs == ('C3', 'C4', 'C5', 'P6', 'P7') and c.backbone.num_features == 128
    assert c.detection.num_depth == 4 and c.detection.groups == 16 and len(c.detection.pr_scales) == 3
    assert (c.detection.min_confidence, c.detection.nms_iou_threshold, c.detection.post_iou_threshold,
            c.detection.nms_max_output_size) == (0.5, 0.4, 0.6, 100)
    assert (c.instance.max_k, c.instance.base_size, tuple(c.instance.crop_size)) == (2, 36, (14, 14))
    assert tuple(c.semantic.atrous_rate) == (6, 12, 18) and c.semantic.num_skip_features == 32
    assert c.train.inference_batch_size == 1 and c.postprocess.resolution == (540, 960)
    d = c.to_dict()
    d["detection"]["groups"] = 8
    c2 = ModelConfiguration()
    c2.from_dict(d)
    assert c2.detection.groups == 8 and ModelConfiguration().detection.groups == 16   # no aliasing
    c2.update("semantic", "num_depth", 2)
    assert c2.semantic.num_depth == 2
    ns = c.get_arg_parser(argv=["-detection.groups", "4", "-instance.crop_size", "7", "7"])
    assert getattr(ns, "detection.groups") == 4 and getattr(ns, "instance.crop_size") == [7, 7]


@pytest.mark.parametrize("bt,ntensors", [("resnext50", 569), ("mobilenet", 439)])
def test_build_api_names_and_weight_specs(bt, ntensors):
    from masklab_hip import ModelConfiguration, retinamasklab as R
    cfg = ModelConfiguration()
    cfg.backbone.backbone_type = bt
    R.K.clear_session()
    bb = R.build_backbone_network(cfg)
    assert bb.output_names == ["C3", "C4", "C5", "P6", "P7"]
    prior, fpn, cls, loc = R.build_detection_network(cfg)
    restore, dist, roi, mask = R.build_instance_network(cfg)
    aspp, seg = R.build_semantic_network(cfg)
    names = [l.name for l in (prior, fpn, cls, loc, restore, dist, roi, mask, aspp, seg)]
    assert names == ["prior_layer", "feature_pyramid", "classification_sub_net", "box_regression_sub_net",
                     "restore_boxes", "mask_distribute", "pyramid_roi_align", "mask_sub_net", "aspp_network",
                     "segmentation_sub_net"]
    model = R.construct_inference_network(cfg, bb, (prior, fpn, cls, loc), (aspp, seg), (restore, dist, roi, mask))
    assert model.output_names == ["cls_pred", "loc_pred", "roi_boxes", "roi_masks", "seg_pred"]
    specs = model.weight_specs()
    assert len(specs) == ntensors
    for must in ["P6_conv/kernel", "P6_norm/gamma", "P7_conv/bias", "feature_pyramid/P3/kernel", "aspp_1x1/kernel",
                 "aspp_6_depthwise/depthwise_kernel", "aspp_18_pointwise_GN/beta", "aspp_pool/kernel",
                 "concat_projection_GN/gamma", "skip_projection/kernel", "skip_projection_GN/gamma",
                 "classification_sub_net/block4/output/bias", "mask_sub_net/block2/deconv/kernel"]:
        assert must in specs, must
    assert specs["mask_sub_net/block0/deconv/kernel"].shape == (2, 2, 128, 128)
    assert specs["classification_sub_net/block0/output/kernel"].shape == (3, 3, 128, 75)
    assert specs["P6_norm/gamma"].shape == (128,)
    w = model.init_weights(0)
    np.testing.assert_allclose(w["classification_sub_net/block0/output/bias"], -np.log(99.0), rtol=1e-6)
    assert np.array_equal(model.init_weights(0)["P6_conv/kernel"], w["P6_conv/kernel"])    # deterministic
    import re as _re
    assert R.find_layer_name(_re.compile("^classification_sub_net*"), model) == ["classification_sub_net"]
    # the reference's regex re-wiring names all resolve (retinamasklab.py:515-586)
    for pat in ["^prior_layer*", "^feature_pyramid*", "^box_regression_sub_net*", "^restore_boxes*",
                "^mask_distribute*", "^pyramid_roi_align*", "^mask_sub_net*", "^aspp*", "^segmentation_sub_net*"]:
        assert len(R.find_layer_name(_re.compile(pat), model)) == 1, pat
    if bt == "resnext50":
        assert specs["conv2_block1_2_conv/depthwise_kernel"].shape == (3, 3, 128, 4)
        assert specs["conv5_block3_3_conv/kernel"].shape == (1, 1, 1024, 2048)
        assert "conv1_bn/moving_variance" in specs
    else:
        assert specs["conv_dw_13/depthwise_kernel"].shape == (3, 3, 1024, 1)
        assert specs["conv1/kernel"].shape == (3, 3, 3, 32)


def test_get_config_mirrors_reference_keys():
    from masklab_hip import get_custom_objects
    from masklab_hip.layers import (ClassificationSubNet, DetectionProposal, FeaturePyramid, MaskDistribute,
                                    PyramidRoiAlign, SegmentationSubNet)
    from masklab_hip import GroupNormalization
    assert set(FeaturePyramid([8, 16, 32], 128).get_config()) >= {"strides", "num_features", "name"}
    cfgd = ClassificationSubNet(5, 5).get_config()
    assert set(cfgd) >= {"num_blocks", "num_classes", "num_depth", "num_features", "num_priors", "use_separable_conv",
                         "expand_ratio", "use_squeeze_excite", "squeeze_ratio", "groups"}
    assert DetectionProposal().get_config()["nms_max_output_size"] == 1000          # reference ctor default
    assert MaskDistribute().get_config() == {"name": "mask_distribute", "trainable": True, "max_k": 2, "base_size": 64} \
        or MaskDistribute().get_config()["base_size"] == 64
    assert PyramidRoiAlign().get_config()["max_batch_size"] == 64
    assert SegmentationSubNet().get_config()["num_skip_features"] == 48
    g = GroupNormalization(groups=16).get_config()
    assert g["groups"] == 16 and g["epsilon"] == 1e-5 and g["axis"] == -1
    reg = get_custom_objects()
    for n in ["RestoreBoxes", "PriorLayer", "FeaturePyramid", "ClassificationSubNet", "BoxRegressionSubNet",
              "MaskSubNet", "DetectionProposal", "MaskDistribute", "PyramidRoiAlign", "ASPPNetwork",
              "SegmentationSubNet", "GroupNormalization", "BackBonePreProcess", "ResizeLike", "MoldBatch"]:
        assert n in reg


def test_error_behaviour_matches_reference():
    from masklab_hip import backbone
    from masklab_hip.layers import PriorLayer
    with pytest.raises(NotImplementedError):
        backbone.load_backbone("not_a_backbone")                       # reference base.py:281-282
    with pytest.raises(ValueError):
        PriorLayer(prior=[1, 2, 3])                                    # reference detection.py:259


def test_forward_golden_is_reproducible_from_seeds(golden_dir):
    """the committed end-to-end vector really is oracle(init_weights(seed), rng(image_seed))"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("mk", os.path.join(golden_dir, "make_forward_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    from oracle import masklab as O
    g = np.load(os.path.join(golden_dir, "forward_mobilenet_128.npz"))
    cfg, _model, w, images = mk.build_case(*mk.CASES["forward_mobilenet_128"])
    cfg.detection.min_confidence = float(g["min_confidence"])
    outs, internals = O.inference_forward(cfg, w, images, literal_groups=False, return_internals=True)
    np.testing.assert_array_equal(internals["kept"], g["kept"])
    for name, o in zip(["cls_pred", "loc_pred", "roi_boxes", "roi_masks", "seg_pred"], outs):
        np.testing.assert_allclose(o, g[name], rtol=1e-5, atol=1e-5)


# ----------------------------------------------------------------------------- deploy wrapper
def test_deploy_layers_config_and_registry():
    import masklab_hip as M
    from masklab_hip import layers as L
    reg = M.get_custom_objects()
    for name in ("DownSampleInput", "UpSampleOutput", "TrimInstances", "SemanticSmoothing", "CropAndPadMask",
                 "CrackToInstance", "SummaryOutput", "IncludeMyRoad", "CalculateInstanceSize"):
        assert reg[name] is getattr(L, name)
    assert L.SummaryOutput().get_config()["default_road_size"] == 3.25             # reference misc.py:567
    assert L.CalculateInstanceSize().get_config()["default_road_size"] == 3.25
    assert L.IncludeMyRoad().get_config()["threshold"] == 0.1 and L.CrackToInstance().get_config()["crack_id"] == 5
    assert L.DownSampleInput().get_config()["target_size"] == (540, 960)          # reference misc.py:141
    assert L.TrimInstances().get_config()["mold"] is True and L.TrimInstances().get_config()["max_batch_size"] == 64
    c = L.SemanticSmoothing().get_config()
    assert c["kernel_size"] == 10 and c["weight"] == 1.0                          # reference semantic.py:263
    d = L.DownSampleInput((540, 960))
    assert d.output_size(1080, 1920) == (540, 960)
    assert d.output_size(720, 960) == (540, 720)            # aspect ratio kept, truncated
    assert d.output_size(400, 640) == (540, 864)            # small inputs are enlarged the same way


def test_deploy_layers_refuse_cpu_tensors():
    import torch
    from masklab_hip import layers as L
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        L.DownSampleInput()(torch.zeros((1, 8, 8, 3), dtype=torch.uint8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        L.SemanticSmoothing(3)(torch.zeros((1, 8, 8, 3)))


class _FakeH5(dict):
    """the slice of the h5py mapping protocol the converter relies on"""

    def __init__(self, *a, attrs=None, **k):
        super().__init__(*a, **k)
        self.attrs = attrs or {}


def _fake_keras_file(weights, nested_prefix=""):
    by_layer = {}
    for name, arr in weights.items():
        by_layer.setdefault(name.split("/")[0], []).append((name, arr))
    root = _FakeH5(attrs={"layer_names": [k.encode() for k in by_layer]})
    for lname, items in by_layer.items():
        grp = _FakeH5(attrs={"weight_names": [(nested_prefix + n + ":0").encode() for n, _ in items]})
        for n, arr in items:
            node = grp
            parts = (nested_prefix + n + ":0").split("/")
            for part in parts[:-1]:
                node = node.setdefault(part, _FakeH5())
            node[parts[-1]] = arr
        root[lname] = grp
    return _FakeH5({"model_weights": root})


def test_keras_h5_converter_walks_and_matches_names():
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "convert_keras_h5", os.path.join(os.path.dirname(__file__), "..", "tools", "convert_keras_h5.py"))
    conv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(conv)
    from masklab_hip import ModelConfiguration, retinamasklab as R
    cfg = ModelConfiguration()
    cfg.backbone.backbone_type = "mobilenet"
    _, model = R.construct_masklab_networks(cfg)
    w = model.init_weights(1)
    specs = {k: tuple(v.shape) for k, v in model.weight_specs().items()}
    # a model.save file, plus an optimizer-ish extra tensor that must be reported as unused
    f = _fake_keras_file(dict(w, **{"training_only_layer/kernel": np.zeros((2, 2), np.float32)}))
    got = conv.collect_h5_weights(f)
    assert set(got) == set(w) | {"training_only_layer/kernel"}
    matched, report = conv.match_to_model(got, specs)
    assert report["missing"] == [] and report["shape_mismatch"] == []
    assert report["unexpected"] == ["training_only_layer/kernel"]
    for k in w:
        np.testing.assert_array_equal(matched[k], w[k])
    # save_weights layout (no model_weights group)
    got2 = conv.collect_h5_weights(f["model_weights"])
    assert set(got2) == set(got)
    # an outer model scope in the file ("inference/<name>") is matched by unique suffix
    scoped, rep = conv.match_to_model({"inference/" + k: v for k, v in w.items()}, specs)
    assert rep["missing"] == [] and set(scoped) == set(w)
    # a missing tensor and a wrong shape are reported, not silently skipped
    bad = dict(got)
    first = sorted(w)[0]
    del bad[first]
    second = sorted(w)[1]
    bad[second] = np.zeros((1,), np.float32)
    _, rep = conv.match_to_model(bad, specs)
    assert rep["missing"] == [first] and rep["shape_mismatch"][0][0] == second


def _reference_named(weights, cfg, uid_start):
    """Re-key a weight dict of THIS package the way the reference's Keras model names the same tensors: sub-layers the
    reference leaves unnamed get Keras' automatic `<class>_N` names, N from one session-wide counter per class in
    CONSTRUCTION order (engine/retinamasklab.py:40-198 build order; constructors detection.py:39-48,109-130,
    instance.py:177-201, semantic.py:199-219), Dense layers of SqueezeExcite at first CALL (misc.py:34-40, call order
    of construct_inference_network :443-486).  Written from the reference constructors, not from the converter."""
    import collections
    uid = collections.defaultdict(int, uid_start)

    def auto(cls):
        n = uid[cls]
        uid[cls] += 1
        return cls if n == 0 else f"{cls}_{n}"

    det, ins, sem = cfg.detection, cfg.instance, cfg.semantic
    pre = {}                                            # our sub-layer prefix -> reference prefix
    se_calls = []                                       # SqueezeExcite instances in call order
    for p in (5, 4, 3):
        pre[f"feature_pyramid/C{p}_lateral"] = "feature_pyramid/" + auto("conv2d")

    def tower(scope, base, depth, se, sep):
        for i in range(depth):
            if se:
                pre[f"{base}/se{i}"] = f"{scope}/" + auto("squeeze_excite")
                se_calls.append(f"{base}/se{i}")
            if sep:
                pre[f"{base}/sep{i}"] = f"{scope}/" + auto("mobile_separable_conv2d")
            else:
                pre[f"{base}/conv{i}"] = f"{scope}/" + auto("conv2d")
            pre[f"{base}/gn{i}"] = f"{scope}/" + auto("group_normalization")

    for scope, se in (("classification_sub_net", det.use_squeeze_excite),
                      ("box_regression_sub_net", det.use_separable_conv)):          # builder quirk :95
        for b in range(5):
            tower(scope, f"{scope}/block{b}", det.num_depth, se, det.use_separable_conv)
            pre[f"{scope}/block{b}/output"] = f"{scope}/" + auto("conv2d")
    for b in range(ins.max_k + 1):
        tower("mask_sub_net", f"mask_sub_net/block{b}", ins.num_depth, ins.use_squeeze_excite, ins.use_separable_conv)
        pre[f"mask_sub_net/block{b}/deconv"] = "mask_sub_net/" + auto("conv2d_transpose")
        pre[f"mask_sub_net/block{b}/output"] = "mask_sub_net/" + auto("conv2d")
    tower("segmentation_sub_net", "segmentation_sub_net", sem.num_depth, sem.use_squeeze_excite, sem.use_separable_conv)
    pre["segmentation_sub_net/output"] = "segmentation_sub_net/" + auto("conv2d")
    dense = {}
    for se in se_calls:                                 # (mask head before the decoder: retinamasklab.py:470 then :486)
        dense[se] = (auto("dense"), auto("dense"))
    out = {}
    for k, v in weights.items():
        parts = k.split("/")
        new = None
        for cut in range(len(parts) - 1, 0, -1):
            head = "/".join(parts[:cut])
            if head in pre:
                new = pre[head] + "/" + "/".join(parts[cut:])
                if head in dense:                       # .../se{i}/dense{1,2}/kernel
                    new = pre[head] + "/" + dense[head][int(parts[cut][5:]) - 1] + "/" + "/".join(parts[cut + 1:])
                break
            m = re.match(r"^(.*/sep\d+)_(.+)$", head)
            if m and m.group(1) in pre and cut == len(parts) - 1:
                new = f"{pre[m.group(1)]}/SeparableConv2d_{m.group(2)}/" + parts[-1]
                break
        if new is None:
            if parts[0].startswith(("aspp_", "concat_projection")):
                new = "aspp_network/" + k               # explicit names inside ASPPNetwork (semantic.py:112-136)
            elif parts[0].startswith("skip_projection"):
                new = "segmentation_sub_net/" + k       # semantic.py:199-201
            else:
                new = k
        assert new not in out, new
        out[new] = v
    return out


@pytest.mark.parametrize("flags,uid_start", [
    ({}, {}),                                                                   # default heads, fresh session
    ({}, {"conv2d": 7, "group_normalization": 3, "conv2d_transpose": 2}),       # session not cleared before the build
    ({"se": True, "sep": True}, {"dense": 5, "squeeze_excite": 1}),             # the optional tower variants
])
def test_keras_h5_converter_maps_reference_auto_names_by_creation_order(flags, uid_start):
    """a REAL checkpoint names tower / lateral / mask-head sub-layers `conv2d_N`,
    `group_normalization_N`, ... (the reference never names them); the converter must map them onto this package's
    hierarchical names by creation order, whatever N the session counters started from."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "convert_keras_h5", os.path.join(os.path.dirname(__file__), "..", "tools", "convert_keras_h5.py"))
    conv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(conv)
    from masklab_hip import ModelConfiguration, retinamasklab as R
    cfg = ModelConfiguration()
    cfg.backbone.backbone_type = "mobilenet"
    if flags.get("se"):
        cfg.detection.use_squeeze_excite = cfg.instance.use_squeeze_excite = cfg.semantic.use_squeeze_excite = True
    if flags.get("sep"):
        cfg.detection.use_separable_conv = cfg.instance.use_separable_conv = True
    _, model = R.construct_masklab_networks(cfg)
    w = model.init_weights(2)
    specs = {k: tuple(v.shape) for k, v in model.weight_specs().items()}
    ref_named = _reference_named(w, cfg, uid_start)
    assert any(re.search(r"/conv2d_\d+/kernel$", k) for k in ref_named)          # the stand-in really is auto-named
    assert not any("/block" in k for k in ref_named)
    got = conv.collect_h5_weights(_fake_keras_file(ref_named))
    assert set(got) == set(ref_named)
    # without the creation-order mapping nothing under the towers is found (what round 1's converter did)
    _, rep0 = conv.match_to_model(got, specs)
    assert len(rep0["missing"]) > 100
    table = []
    matched, rep = conv.match_to_model(conv.rename_keras_auto_names(got, specs, table), specs)
    assert rep["missing"] == [] and rep["shape_mismatch"] == [] and rep["unexpected"] == []
    # the applied creation-order pairing is reported row by row (what a user with a real .h5 audits)
    assert len(table) == len({r[4] for r in table}) and all(r[3].split("/")[-1].startswith(r[1]) for r in table)
    laterals = [r for r in table if r[0] == "feature_pyramid"]
    assert [r[4] for r in sorted(laterals, key=lambda r: r[2])] == [f"feature_pyramid/C{k}_lateral" for k in (5, 4, 3)]
    import io
    buf = io.StringIO()
    conv.print_order_table(table, file=buf)
    assert buf.getvalue().count("  ->  ") == len(table)
    for k in w:
        np.testing.assert_array_equal(matched[k], w[k], err_msg=k)
    # a checkpoint of a different head configuration is refused, not mis-assigned
    fewer = {k: v for k, v in got.items() if "/group_normalization" not in k or not k.startswith("mask_sub_net")}
    dropped = sorted(k for k in got if k.startswith("mask_sub_net/group_normalization"))[:2]
    fewer = {k: v for k, v in got.items() if k not in dropped}
    with pytest.raises(ValueError, match="different head configuration"):
        conv.rename_keras_auto_names(fewer, specs)


def test_out1x1_lane_table_reproduces_the_1x1_conv():
    """packing.pack_out1x1_table (operand of ml_deconv2x2_out1x1_f32): emulate what the kernel does with it -- the
    transposed product leaves, in lane l / register e of channel tile t, channel 32t + (e&3) + 8(e>>2) + 4(l>>5) of
    pixel l&31; one MFMA per register then contracts the two lane halves against table[t][e][half][class] -- and
    compare with the plain 1x1 conv."""
    from masklab_hip import packing
    rng = np.random.default_rng(5)
    for cmid, ncls in [(128, 3), (256, 1), (128, 20), (128, 32)]:
        k = rng.normal(size=(1, 1, cmid, ncls)).astype(np.float32)
        b = rng.normal(size=(ncls,)).astype(np.float32)
        table, bo, cp = packing.pack_out1x1_table(k, b)
        assert cp >= ncls and cp & (cp - 1) == 0 and cp < 2 * max(ncls, 1) + 1
        assert table.shape == (cmid // 32, 16, 2, cp) and bo.shape == (cp,)
        assert not table[..., ncls:].any() and not bo[ncls:].any()
        t_act = rng.normal(size=(32, cmid))                      # [pixel, channel]: the activated transposed-conv tile
        y = np.tile(bo.astype(np.float64), (32, 1))
        for t in range(cmid // 32):
            for e in range(16):
                for half in range(2):                            # the MFMA's k index = lane half
                    ch = 32 * t + (e & 3) + 8 * (e >> 2) + 4 * half
                    y += np.outer(t_act[:, ch], table[t, e, half].astype(np.float64))
        want = t_act @ k[0, 0].astype(np.float64) + b
        np.testing.assert_allclose(y[:, :ncls], want, atol=1e-12)
    with pytest.raises(ValueError):
        packing.pack_out1x1_table(np.zeros((1, 1, 128, 33), np.float32))
    with pytest.raises(ValueError):
        packing.pack_out1x1_table(np.zeros((1, 1, 100, 3), np.float32))


def test_small_launches_pick_narrow_tiles():
    """ml_conv2d_launch_ntile (host logic only, no GPU needed): a launch whose 128-wide tiles x K splits cannot fill the
    resident blocks runs on 128x32 (x4 <= 512) or 128x64 (x2 <= 512) tiles; big launches, grouped convs and launches
    packed for 32-wide tiles keep theirs.  The split-K count is taken from the 128-wide tile count either way."""
    from masklab_hip import _lib
    lib = _lib.load()

    def desc(B, H, W, cin, cout, k=3, group_step=0, tile=0):
        d = _lib.ConvDesc()
        d.B, d.H, d.W, d.Ho, d.Wo = B, H, W, H, W
        d.in_cstride, d.in_coff, d.span, d.span_pad, d.cpp_shift = cin, 0, cin, -(-cin // 32) * 32, 30
        d.KH = d.KW = k
        d.stride = d.dil = 1
        d.cout, d.n_pad, d.out_cstride = cout, -(-cout // 128) * 128, cout
        d.group_cin_step, d.tile, d.math = group_step, tile, 0
        return d

    def ntile(ds, ws=1):
        arr = (_lib.ConvDesc * len(ds))(*ds)
        return lib.ml_conv2d_launch_ntile(arr, len(ds), ws)

    assert ntile([desc(8, 128, 128, 128, 128)]) == 128                 # 1024 tiles: fills the chip
    assert ntile([desc(1, 64, 64, 128, 128)]) == 32                    # 32 tiles x 4 splits = 128 blocks -> x4 = 512
    assert ntile([desc(1, 64, 64, 128, 128)], ws=0) == 32              # no workspace: no split-K, 32 tiles
    assert ntile([desc(2, 64, 64, 128, 128)]) == 64                    # 64 tiles x 4 splits = 256 blocks: x2 fits, x4 not
    assert ntile([desc(4, 64, 64, 128, 128)]) == 128                   # 128 tiles x 4 splits = 512 blocks
    levels = [desc(1, s, s, 128, 128) for s in (64, 32, 16, 8, 4)]     # the five tower levels of one 512x512 image
    assert ntile(levels) == 64                                         # 44 tiles x 4 splits = 176 blocks: x2 fits, x4 not
    assert ntile([desc(1, 16, 16, 1024, 1024, k=3, group_step=32, tile=3)]) == 32    # grouped: packed for 32 already
    assert ntile([desc(1, 64, 64, 128, 48)]) == 32                     # cout <= 96 picks 32 / 64 by itself
    assert lib.ml_conv2d_launch_ntile(None, 1, 1) == 0


def test_graft_entry_build_runs():
    """`__graft_entry__.build()` is the driver's "does it build" check: make (up to date -> a no-op), import the package,
    bind every symbol, header / library / binding ABI versions agree."""
    import importlib
    g = importlib.import_module("__graft_entry__")
    g.build()
    from masklab_hip import _lib
    src = open(os.path.join(os.path.dirname(__file__), "..", "include", "masklab_hip.h")).read()
    assert f"#define ML_ABI_VERSION {_lib.ABI_VERSION} " in src
