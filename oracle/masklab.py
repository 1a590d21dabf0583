"""CPU ORACLE (test infrastructure only) -- the MaskLab inference forward in NumPy.

PARITY UNPINNED (see oracle/tfops.py header): a restatement of the reference's
layer code, written from the reference text; the reference itself cannot run here
(TensorFlow absent) and ships no fixtures for this path.  Only the anchor table is
pinned against the reference (tests/golden/prior_tables.npz).

Each function cites the reference file:line it follows.  `weights` is a flat dict
{"<layer name>/<weight name>": ndarray} in Keras layouts:
Conv2D kernel [kh,kw,cin,cout] (+bias), DepthwiseConv2D depthwise_kernel
[kh,kw,cin,mult], Conv2DTranspose kernel [kh,kw,cout,cin], BN
gamma/beta/moving_mean/moving_variance, GN gamma/beta.  Layer names are the
reference's explicit names where it gives them and a deterministic hierarchical
name (`<layer>/block<i>/conv<j>` ...) where Keras would auto-number.

`config` is any object exposing the reference's ModelConfiguration attribute tree
(engine/config.py:47-116).  The oracle never imports the product package.
"""
import numpy as np

from . import tfops as T

F32 = np.float32


# =========================================================================== anchors
def prior_table(strides, sizes, pr_scales, pr_ratios):
    """engine/prior.py:55-67: rows (stride, w, h) looping size/stride -> scale -> ratio,
    w=round(size*s*sqrt(r)), h=round(size*s/sqrt(r)) (np.round = half-to-even)."""
    rows = []
    for size, stride in zip(sizes, strides):
        for s in pr_scales:
            for r in pr_ratios:
                w = int(np.round(size * s * np.sqrt(r)))
                h = int(np.round(size * s / np.sqrt(r)))
                rows.append((stride, w, h))
    return np.asarray(rows, np.int64).reshape(-1, 3)


def prior_boxes(table, height, width, padding="same"):
    """PriorLayer.call, engine/layers/detection.py:269-298 (without the batch tile):
    per stride group ascending (:260-262); per row: centres range(stride//2,
    ceil(H/stride)*stride, stride) (:276-284); meshgrid(xs,ys); rows stacked on axis 2
    -> [Hl,Wl,n,4] -> (-1,4) (:289-293); levels concatenated (:295)."""
    out = []
    for stride in sorted(set(table[:, 0].tolist())):
        rows = table[table[:, 0] == stride]
        boxes = []
        for _, bw, bh in rows:
            if padding == "same":
                th = int(np.ceil(height / stride) * stride)
                tw = int(np.ceil(width / stride) * stride)
            else:
                th = int(np.floor(height / stride) * stride)
                tw = int(np.floor(width / stride) * stride)
            ys = np.arange(stride // 2, th, stride)
            xs = np.arange(stride // 2, tw, stride)
            xs, ys = np.meshgrid(xs, ys)
            boxes.append(np.stack((xs, ys, np.ones_like(xs) * bw, np.ones_like(ys) * bh), axis=-1))
        boxes = np.stack(boxes, axis=2).reshape(-1, 4)
        out.append(boxes)
    return np.concatenate(out, axis=0).astype(np.int32)


# =========================================================================== backbone
def backbone_preprocess(x, rgb=True, mean_shift=False, normalize=0):
    """BackBonePreProcess.call, engine/backbone/base.py:57-75."""
    dt = x.dtype
    if rgb:
        mean = np.asarray([123.68, 116.779, 103.939], dt)
        std = np.asarray([0.225, 0.224, 0.229], dt)
    else:
        mean = np.asarray([103.939, 116.779, 123.68], dt)
        std = np.asarray([0.229, 0.224, 0.225], dt)
    if not rgb:
        x = x[..., ::-1]
    if mean_shift:
        x = x - mean
    if normalize == 1:
        return x / dt.type(255.)
    if normalize == 2:
        return x / dt.type(127.5) if mean_shift else x / dt.type(127.5) - dt.type(1.)
    if normalize == 3:
        return (x / dt.type(255.)) / std
    return x


def _bn(x, w, name, eps):
    return T.batch_norm(x, w.get(name + "/gamma"), w[name + "/beta"],
                        w[name + "/moving_mean"], w[name + "/moving_variance"], eps)


def grouped_conv_literal(x, dw_kernel, groups, c, stride):
    """ResNeXt grouped 3x3 exactly as the reference spells it (ResNext.py:212-219):
    ZeroPadding2D(1) + DepthwiseConv2D(depth_multiplier=c, 'valid') (:213-215), then
    SplitGroups reshape [...,groups,c,c] (:33-37), ReduceGroups reduce_sum(axis=-2)
    (:54), MergeGroups reshape [...,filters] (:66-71)."""
    y = T.depthwise_conv2d(x, dw_kernel, stride=stride, padding=((1, 1), (1, 1)))
    B, H, W, _ = y.shape
    y = y.reshape(B, H, W, groups, c, c).sum(axis=-2)
    return y.reshape(B, H, W, groups * c)


def grouped_conv_fast(x, dw_kernel, groups, c, stride):
    """Same arithmetic as grouped_conv_literal without the c-times-larger temporary:
    out[g*c+m] = sum_i conv(x[g*c+i], K[..,g*c+i,m]).  Cross-checked
    against the literal form in tests; used for large inputs / CPU timing."""
    B, H, W, C = x.shape
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
    Ho = (H + 2 - 3) // stride + 1
    Wo = (W + 2 - 3) // stride + 1
    k = dw_kernel.astype(x.dtype).reshape(3, 3, groups, c, c)  # [kh,kw,g,i,m]
    out = np.zeros((B, Ho, Wo, groups, c), x.dtype)
    for i in range(3):
        for j in range(3):
            xs = xp[:, i:i + (Ho - 1) * stride + 1:stride, j:j + (Wo - 1) * stride + 1:stride, :]
            xs = xs.reshape(B, Ho, Wo, groups, c)
            out += np.einsum("bhwgi,gim->bhwgm", xs, k[i, j], optimize=True)
    return out.reshape(B, Ho, Wo, C)


def resnext50(x, w, literal_groups=True):
    """ResNeXt50, engine/backbone/ResNext.py:180-253,343-354,399-416.  Returns dict of
    the C1..C5 taps named in BACKBONE_LAYERS['resnext50'] (base.py:147-153)."""
    eps = 1.001e-5
    gconv = grouped_conv_literal if literal_groups else grouped_conv_fast
    taps = {}
    x = T.conv2d(x, w["conv1_conv/kernel"], None, stride=2, padding=((3, 3), (3, 3)))  # :343-344
    x = T.relu(_bn(x, w, "conv1_bn", eps))                                            # :347-349
    taps["C1"] = x
    x = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))                                   # :351
    x = T.max_pool(x, 3, 2)                                                           # :352

    def block(x, filters, stride, conv_shortcut, name, groups=32):
        if conv_shortcut:                                                             # :199-203
            sc = T.conv2d(x, w[name + "_0_conv/kernel"], None, stride=stride, padding="valid")
            sc = _bn(sc, w, name + "_0_bn", eps)
        else:
            sc = x
        y = T.conv2d(x, w[name + "_1_conv/kernel"], None, padding="valid")            # :207
        y = T.relu(_bn(y, w, name + "_1_bn", eps))
        c = filters // groups
        y = gconv(y, w[name + "_2_conv/depthwise_kernel"], groups, c, stride)         # :212-219
        y = T.relu(_bn(y, w, name + "_2_bn", eps))
        y = T.conv2d(y, w[name + "_3_conv/kernel"], None, padding="valid")            # :225
        y = _bn(y, w, name + "_3_bn", eps)
        return T.relu(sc + y)                                                         # :230-231

    def stack(x, filters, blocks, stride1, name):                                     # :235-253
        x = block(x, filters, stride1, True, name + "_block1")
        for i in range(2, blocks + 1):
            x = block(x, filters, 1, False, name + "_block" + str(i))
        return x

    x = stack(x, 128, 3, 1, "conv2"); taps["C2"] = x                                  # :407-410
    x = stack(x, 256, 4, 2, "conv3"); taps["C3"] = x
    x = stack(x, 512, 6, 2, "conv4"); taps["C4"] = x
    x = stack(x, 1024, 3, 2, "conv5"); taps["C5"] = x
    return taps


def group_conv2d_sliced(x, k, stride):
    """thirdparty GroupConv2D (_common_blocks.py:13-76): per group slice -> Conv2D('valid') -> concat.
    k: [groups,3,3,c,c]; the ZeroPadding2D(1) that precedes it (resnext.py:84,122) is applied here."""
    groups, c = k.shape[0], k.shape[3]
    outs = []
    for g in range(groups):
        outs.append(T.conv2d(x[..., g * c:(g + 1) * c], k[g], None, stride=stride, padding=((1, 1), (1, 1))))
    return np.concatenate(outs, axis=-1)


def resnext101(x, w, repetitions=(3, 4, 23, 3)):
    """thirdparty/classification_models/models/resnext.py:138-241 (ResNeXt101 :266-275), BN eps 2e-5 (:49).
    Build-side extension: the reference never calls it."""
    eps = 2e-5
    taps = {}
    x = T.batch_norm(x, None, w["bn_data/beta"], w["bn_data/moving_mean"], w["bn_data/moving_variance"], eps)  # :194
    x = T.conv2d(x, w["conv0/kernel"], None, stride=2, padding=((3, 3), (3, 3)))                               # :195-196
    x = T.relu(_bn(x, w, "bn0", eps))
    taps["C1"] = x
    x = T.max_pool(np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0))), 3, 2)                                          # :199-200
    for stage, rep in enumerate(repetitions):
        for block in range(rep):
            base = f"stage{stage + 1}_unit{block + 1}_"
            stride = 1 if (stage == 0 or block > 0) else 2                                                    # :209-216
            y = T.relu(_bn(T.conv2d(x, w[base + "conv1/kernel"], None, padding="valid"), w, base + "bn1", eps))
            y = T.relu(_bn(group_conv2d_sliced(y, w[base + "conv2/kernel"], stride), w, base + "bn2", eps))
            y = _bn(T.conv2d(y, w[base + "conv3/kernel"], None, padding="valid"), w, base + "bn3", eps)
            if block == 0:                                                                                     # conv_block :93-95
                sc = _bn(T.conv2d(x, w[base + "sc/kernel"], None, stride=stride, padding="valid"), w, base + "sc_bn", eps)
            else:
                sc = x
            x = T.relu(y + sc)
        taps[f"C{stage + 2}"] = x
    return taps


MOBILENET_BLOCKS = [(64, 1), (128, 2), (128, 1), (256, 2), (256, 1), (512, 2)] + \
    [(512, 1)] * 5 + [(1024, 2), (1024, 1)]


def mobilenet_v1(x, w):
    """tf.keras.applications.MobileNet(alpha=1.0, include_top=False) -- NOT in the
    reference tree (call site base.py:253-258, taps :161-167); architecture restated
    from keras-applications mobilenet.py (parity unpinned):
    conv1: pad ((0,1),(0,1)) + 3x3 s2 valid, no bias, BN(eps 1e-3), ReLU6; 13 blocks of
    [dw 3x3 (s2: same one-sided pad + valid; s1: same) + BN + ReLU6 + 1x1 + BN + ReLU6]."""
    eps = 1e-3
    taps = {}
    x = T.conv2d(x, w["conv1/kernel"], None, stride=2, padding=((0, 1), (0, 1)))
    x = T.relu6(_bn(x, w, "conv1_bn", eps))
    tapname = {1: "C1", 3: "C2", 5: "C3", 11: "C4", 13: "C5"}
    for i, (filters, stride) in enumerate(MOBILENET_BLOCKS, start=1):
        pad = "same" if stride == 1 else ((0, 1), (0, 1))
        x = T.depthwise_conv2d(x, w["conv_dw_%d/depthwise_kernel" % i], stride=stride, padding=pad)
        x = T.relu6(_bn(x, w, "conv_dw_%d_bn" % i, eps))
        x = T.conv2d(x, w["conv_pw_%d/kernel" % i], None, padding="same")
        x = T.relu6(_bn(x, w, "conv_pw_%d_bn" % i, eps))
        if i in tapname:
            taps[tapname[i]] = x
    return taps


PREPROCESS = {  # base.py:190-279
    "resnext50": dict(rgb=True, mean_shift=True, normalize=2),
    "mobilenet": dict(rgb=False, mean_shift=False, normalize=2),
    "resnext101": dict(rgb=True, mean_shift=False, normalize=0),     # extension: raw RGB into bn_data
}


def backbone_forward(images, w, backbone_type, backbone_outputs, literal_groups=True):
    """load_backbone graph, engine/backbone/base.py:185-316.  Returns (names, tensors)
    in the model's output order: C-taps ascending, then P6, P7."""
    bt = backbone_type.lower()
    if bt not in PREPROCESS:
        raise NotImplementedError(bt)
    x = backbone_preprocess(images, **PREPROCESS[bt])
    if bt == "resnext50":
        taps = resnext50(x, w, literal_groups)
    elif bt == "resnext101":
        taps = resnext101(x, w)
    else:
        taps = mobilenet_v1(x, w)
    names, feats = [], []
    for key in ("C1", "C2", "C3", "C4", "C5"):                                        # :287-290
        if key in backbone_outputs:
            names.append(key); feats.append(taps[key])
    last = feats[-1]
    pad = ((0, 1), (0, 1)) if bt == "mobilenet" else "same"                           # :292-314
    p6 = T.relu(T.conv2d(last, w["P6_conv/kernel"], w["P6_conv/bias"], stride=2, padding=pad))
    if "P6" in backbone_outputs:
        names.append("P6"); feats.append(p6)                                          # pre-norm
    g6 = T.group_norm(p6, w["P6_norm/gamma"], w["P6_norm/beta"], 32)                  # default groups
    p7 = T.relu(T.conv2d(g6, w["P7_conv/kernel"], w["P7_conv/bias"], stride=2, padding=pad))
    if "P7" in backbone_outputs:
        names.append("P7"); feats.append(p7)
    return names, feats


# =========================================================================== detection head
def feature_pyramid(inputs, w, strides, prefix="feature_pyramid"):
    """FeaturePyramid.call, detection.py:51-66: top-down over inputs[::-1]; lateral 1x1
    (:56); upsample prev (pre-3x3 sum) to lateral size + add (:58-60); 3x3 'P{k}' (:64)."""
    prev = None
    outs = []
    for stride, head in zip(sorted(strides, reverse=True), inputs[::-1]):
        p = int(np.round(np.log2(stride)))
        lat = T.conv2d(head, w[f"{prefix}/C{p}_lateral/kernel"], w[f"{prefix}/C{p}_lateral/bias"])
        if prev is not None:
            up = T.resize_bilinear_align_corners(prev, lat.shape[1], lat.shape[2])
            lat = lat + up
        prev = lat
        outs.append(T.conv2d(lat, w[f"{prefix}/P{p}/kernel"], w[f"{prefix}/P{p}/bias"]))
    return outs[::-1]


def squeeze_excite(x, w, name):
    """SqueezeExcite.call, misc.py:42-47: GAP -> Dense(relu,no bias) -> Dense(sigmoid) -> scale."""
    se = x.mean(axis=(1, 2), dtype=x.dtype)
    se = T.relu(se @ w[name + "/dense1/kernel"].astype(x.dtype))
    se = T.sigmoid(se @ w[name + "/dense2/kernel"].astype(x.dtype))
    return x * se[:, None, None, :]


def mobile_separable_conv2d(x, w, name, groups=16, stride=1):
    """MobileSeparableConv2D.call, misc.py:94-106: 1x1 expand (no bias) -> GN -> ReLU -> depthwise 3x3
    'same' (no bias) -> GN -> ReLU -> 1x1 squeeze (no bias) -> GN -> inputs + x."""
    y = T.conv2d(x, w[name + "_expand_conv/kernel"])
    y = T.relu(_gn(y, w, name + "_expand_GN", groups))
    y = T.depthwise_conv2d(y, w[name + "_depthwise/depthwise_kernel"], stride=stride, padding="same")
    y = T.relu(_gn(y, w, name + "_depthwise_GN", groups))
    y = T.conv2d(y, w[name + "_squeeze_conv/kernel"])
    y = _gn(y, w, name + "_squeeze_GN", groups)
    return x + y


def _tower(x, w, prefix, depth, groups, use_se, use_sep=False):
    """depth x [SqueezeExcite? ; Conv3x3+ReLU | MobileSeparableConv2D ; GN]
    (detection.py:111-125,181-195; instance.py:179-193; semantic.py:205-217)."""
    for i in range(depth):
        if use_se:
            x = squeeze_excite(x, w, f"{prefix}/se{i}")
        if use_sep:
            x = mobile_separable_conv2d(x, w, f"{prefix}/sep{i}")
        else:
            x = T.relu(T.conv2d(x, w[f"{prefix}/conv{i}/kernel"], w[f"{prefix}/conv{i}/bias"]))
        x = T.group_norm(x, w[f"{prefix}/gn{i}/gamma"], w[f"{prefix}/gn{i}/beta"], groups)
    return x


def classification_subnet(features, w, num_classes, depth, groups, use_se=False, use_sep=False,
                          prefix="classification_sub_net"):
    """ClassificationSubNet.call, detection.py:204-212 (ctor :162-202): per level own
    block: depth x [conv3x3+ReLU (:190) ; GN (:194)] ; conv3x3 -> priors*classes + sigmoid
    (:197-200) ; Reshape((-1,nc)) ; concat axis 1."""
    heads = []
    for l, x in enumerate(features):
        p = f"{prefix}/block{l}"
        x = _tower(x, w, p, depth, groups, use_se, use_sep)
        x = T.sigmoid(T.conv2d(x, w[p + "/output/kernel"], w[p + "/output/bias"]))
        heads.append(x.reshape(x.shape[0], -1, num_classes))
    return np.concatenate(heads, axis=1)


def box_regression_subnet(features, w, depth, groups, use_se=False, use_sep=False,
                          prefix="box_regression_sub_net"):
    """BoxRegressionSubNet.call, detection.py:132-140 (ctor :93-130)."""
    heads = []
    for l, x in enumerate(features):
        p = f"{prefix}/block{l}"
        x = _tower(x, w, p, depth, groups, use_se, use_sep)
        x = T.conv2d(x, w[p + "/output/kernel"], w[p + "/output/bias"])
        heads.append(x.reshape(x.shape[0], -1, 4))
    return np.concatenate(heads, axis=1)


def restore_boxes(loc_pred, pr_boxes):
    """RestoreBoxes.call, detection.py:325-344 (float32)."""
    loc = loc_pred.astype(F32)
    pr = pr_boxes.astype(F32)
    cx = loc[..., 0] * pr[..., 2] + pr[..., 0]
    cy = loc[..., 1] * pr[..., 3] + pr[..., 1]
    bw = np.exp(loc[..., 2]) * pr[..., 2]
    bh = np.exp(loc[..., 3]) * pr[..., 3]
    return np.stack([cx, cy, bw, bh], axis=-1).astype(F32)


def normalize_boxes(boxes, shape=(1.0, 1.0)):
    """NormalizeBoxes.call, detection.py:360-375: (cx,cy,w,h) -> (y1,x1,y2,x2)/(H,W);
    default shape = ones => pixel corners (:362)."""
    ih, iw = F32(shape[0]), F32(shape[1])
    b = boxes.astype(F32)
    cx, cy, bw, bh = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    two = F32(2)
    x1 = (cx - bw / two) / iw
    y1 = (cy - bh / two) / ih
    x2 = (cx + bw / two) / iw
    y2 = (cy + bh / two) / ih
    return np.stack([y1, x1, y2, x2], axis=-1).astype(F32)


def detection_proposal(cls_pred, boxes, min_confidence, nms_iou_threshold, post_iou_threshold,
                       nms_max_output_size, max_batch_size=1):
    """DetectionProposal.call, detection.py:482-567 (+ MoldBatch misc.py:231-286).
    Returns (molded [B,N,6], flat kept indices [n,3]=(b,a,c) in output order)."""
    cls_pred = cls_pred.astype(F32)
    boxes = boxes.astype(F32)
    B, A, C = cls_pred.shape
    norm = normalize_boxes(boxes)                                   # :488 (no shape => /1)
    keep = np.argwhere(cls_pred >= F32(min_confidence)).astype(np.int64)   # :491 row-major
    conf = cls_pred[keep[:, 0], keep[:, 1], keep[:, 2]]
    kb = norm[keep[:, 0], keep[:, 1]]
    ids = keep[:, 0] * (C + 1) + keep[:, 2]                         # :519
    _, first = np.unique(ids, return_index=True)
    uniq = ids[np.sort(first)]                                      # tf.unique: first-occurrence order
    per_class = []
    for u in uniq:                                                  # :522 map_fn (sequential)
        ixs = np.nonzero(ids == u)[0]                               # :506
        sel = T.non_max_suppression(kb[ixs], conf[ixs], nms_max_output_size, nms_iou_threshold)
        per_class.append(keep[ixs[sel]])                            # :511
    per_class = np.concatenate(per_class, 0) if per_class else np.zeros((0, 3), np.int64)
    pc_conf = cls_pred[per_class[:, 0], per_class[:, 1], per_class[:, 2]]
    pc_box = norm[per_class[:, 0], per_class[:, 1]]
    final = []
    for b in range(B):                                              # :552-553
        ixs = np.nonzero(per_class[:, 0] == b)[0]                   # :541
        sel = T.non_max_suppression(pc_box[ixs], pc_conf[ixs], nms_max_output_size, post_iou_threshold)
        final.append(per_class[ixs[sel]])
    final = np.concatenate(final, 0) if final else np.zeros((0, 3), np.int64)
    rows = np.concatenate([boxes[final[:, 0], final[:, 1]],                         # :557-563
                           final[:, 2:3].astype(F32),
                           cls_pred[final[:, 0], final[:, 1], final[:, 2]][:, None]], axis=1)
    molded = T.mold_batch(rows.astype(F32), final[:, 0], B)
    return molded, final


# =========================================================================== instance head
def mask_distribute(proposed, max_k=2, base_size=36):
    """MaskDistribute.call, instance.py:52-66 (K.epsilon()=1e-7, float32)."""
    p = proposed.astype(F32)
    eps = F32(1e-7)
    size = np.sqrt(p[..., 2] * p[..., 3])
    with np.errstate(invalid="ignore", divide="ignore"):
        dk = np.log((size + eps) / (F32(base_size) + eps)) / np.log(F32(2.))
    k = np.clip(np.floor(dk), 0, max_k).astype(F32)
    k = np.where(p[..., 0] == F32(-1.), p[..., 0], k)
    return np.concatenate([k[..., None], p], axis=-1)


def pyramid_roi_align(fmaps, dist_boxes, image_hw, crop_size=(14, 14)):
    """PyramidRoiAlign.call, instance.py:109-139: boxes normalised by IMAGE (H,W)
    (:115-116); per level: idx=where(k==level) row-major (:121); crop_and_resize (:125);
    MoldBatch both crops and boxes with -1 (:127-134); roi_boxes = concat axis 1 (:135-138)."""
    B = dist_boxes.shape[0]
    norm = normalize_boxes(dist_boxes[..., 1:5], shape=image_hw)
    roi_fmaps, roi_boxes = [], []
    for level, fmap in enumerate(fmaps):
        idx = np.argwhere(dist_boxes[..., 0] == level)
        tb = norm[idx[:, 0], idx[:, 1]]
        bi = idx[:, 0].astype(np.int32)
        crops = T.crop_and_resize(fmap, tb, bi, crop_size)
        roi_fmaps.append(T.mold_batch(crops, bi, B))
        roi_boxes.append(T.mold_batch(dist_boxes[idx[:, 0], idx[:, 1], 1:], bi, B))
    roi_boxes = np.concatenate(roi_boxes, axis=1) if len(fmaps) > 1 else roi_boxes[0]
    return roi_fmaps, roi_boxes


def mask_subnet(roi_fmaps, w, depth, groups, use_se=False, use_sep=False, prefix="mask_sub_net"):
    """MaskSubNet.call, instance.py:203-225 (ctor :162-201)."""
    heads = []
    for k, x in enumerate(roi_fmaps):
        B, n = x.shape[:2]
        p = f"{prefix}/block{k}"
        x = x.reshape((B * n,) + x.shape[2:])
        x = _tower(x, w, p, depth, groups, use_se, use_sep)
        x = T.relu(T.conv2d_transpose_2x2_s2(x, w[p + "/deconv/kernel"], w[p + "/deconv/bias"]))
        x = T.sigmoid(T.conv2d(x, w[p + "/output/kernel"], w[p + "/output/bias"]))
        heads.append(x.reshape((B, n) + x.shape[1:]))
    return np.concatenate(heads, axis=1) if len(heads) > 1 else heads[0]


# =========================================================================== semantic head
def _gn(x, w, name, groups):
    return T.group_norm(x, w[name + "/gamma"], w[name + "/beta"], groups)


def aspp_network(x, w, atrous_rate=(6, 12, 18), groups=16):
    """ASPPNetwork.call, semantic.py:139-159; AtrousSeparableConv2D.call :75-83."""
    b1 = T.relu(_gn(T.conv2d(x, w["aspp_1x1/kernel"]), w, "aspp_1x1_GN", groups))      # :112-116
    branches = [b1]
    for r in atrous_rate:                                                               # :63-83
        y = T.depthwise_conv2d(x, w[f"aspp_{r}_depthwise/depthwise_kernel"], dilation=r)
        y = T.relu(_gn(y, w, f"aspp_{r}_depthwise_GN", groups))
        y = T.conv2d(y, w[f"aspp_{r}_pointwise/kernel"])
        y = T.relu(_gn(y, w, f"aspp_{r}_pointwise_GN", groups))
        branches.append(y)
    pool = x.mean(axis=(1, 2), keepdims=True, dtype=x.dtype)                            # :149
    pool = T.relu(T.conv2d(pool, w["aspp_pool/kernel"]))                                # :126-129 (no GN)
    pool = T.resize_bilinear_align_corners(pool, x.shape[1], x.shape[2])                # :152
    cat = np.concatenate(branches + [pool], axis=-1)                                    # :154
    y = T.conv2d(cat, w["concat_projection/kernel"])
    return T.relu(_gn(y, w, "concat_projection_GN", groups))                            # :156-157


def segmentation_subnet(aspp_out, skip, w, depth, groups, use_se=False, use_sep=False,
                        prefix="segmentation_sub_net"):
    """SegmentationSubNet.call, semantic.py:221-231 (ctor :183-219)."""
    s = T.relu(_gn(T.conv2d(skip, w["skip_projection/kernel"]), w, "skip_projection_GN", groups))
    up = T.resize_bilinear_align_corners(aspp_out, s.shape[1], s.shape[2])              # :226
    x = np.concatenate([up, s], axis=-1)                                                # :227
    x = _tower(x, w, prefix, depth, groups, use_se, use_sep)
    return T.sigmoid(T.conv2d(x, w[prefix + "/output/kernel"], w[prefix + "/output/bias"]))


# =========================================================================== whole path
def inference_forward(config, weights, images, dtype=np.float32, literal_groups=True,
                      with_detection=True, with_instance=True, with_semantic=True,
                      return_internals=False, min_confidence=None):
    """construct_inference_network, engine/retinamasklab.py:420-495.
    Returns [cls_pred, loc_pred, roi_boxes, roi_masks, seg_pred] (subset by flags).
    min_confidence: None = config.detection.min_confidence (:459-466); a float overrides it; a callable
    cls_pred -> float chooses it from the scores (fixtures put it in a score gap), reported in internals."""
    w = weights
    x = np.asarray(images).astype(dtype)
    B, H, W, _ = x.shape
    names, feats = backbone_forward(x, w, config.backbone.backbone_type,
                                    config.backbone.backbone_outputs, literal_groups)
    outs, internals = [], {"backbone": dict(zip(names, feats))}
    if with_detection:
        det = config.detection
        num_classes = len(config.dataset.instance_labels)
        strides = [2 ** int(n[-1]) for n in config.backbone.backbone_outputs]           # :46-48
        table = prior_table(strides, [4 * s for s in strides], det.pr_scales, det.pr_ratios)
        pr = prior_boxes(table, H, W)                                                   # :434
        fpn_in = [f for n, f in zip(names, feats) if n in det.feature_pyramid_inputs]   # :437-442
        rest = [f for n, f in zip(names, feats) if n not in det.feature_pyramid_inputs]
        fpn_strides = [2 ** int(n[1]) for n in det.feature_pyramid_inputs]
        fouts = feature_pyramid(fpn_in, w, fpn_strides) + rest                          # :443-444
        internals["features"] = fouts
        cls_pred = classification_subnet(fouts, w, num_classes, det.num_depth, det.groups,
                                         det.use_squeeze_excite, det.use_separable_conv)
        loc_pred = box_regression_subnet(fouts, w, det.num_depth, det.groups,
                                         det.use_separable_conv,   # builder quirk :95 (SE flag)
                                         det.use_separable_conv)
        outs += [cls_pred, loc_pred]
        if with_instance:
            ins = config.instance
            boxes = restore_boxes(loc_pred, pr[None])                                   # :458
            thr = det.min_confidence if min_confidence is None else (
                min_confidence(cls_pred) if callable(min_confidence) else min_confidence)
            internals["min_confidence"] = float(thr)
            proposed, kept = detection_proposal(
                cls_pred, boxes, thr, det.nms_iou_threshold,
                det.post_iou_threshold, det.nms_max_output_size,
                config.train.inference_batch_size)                                      # :459-466
            dist = mask_distribute(proposed, ins.max_k, ins.base_size)                  # :467
            roi_fmaps, roi_boxes = pyramid_roi_align(fouts[:ins.max_k + 1], dist, (H, W),
                                                     tuple(ins.crop_size))              # :468-469
            roi_fmaps = [f.astype(dtype) for f in roi_fmaps]
            roi_masks = mask_subnet(roi_fmaps, w, ins.num_depth, ins.groups,
                                    ins.use_squeeze_excite, ins.use_separable_conv)     # :470
            outs += [roi_boxes, roi_masks]
            internals.update(boxes=boxes, proposed=proposed, kept=kept, dist=dist,
                             roi_fmaps=roi_fmaps)
    if with_semantic:
        sem = config.semantic
        fmap = dict(zip(names, feats))
        aspp = aspp_network(fmap[sem.aspp_input_name], w, tuple(sem.atrous_rate), sem.atrous_groups)
        seg = segmentation_subnet(aspp, fmap[sem.skip_input_name], w, sem.num_depth, sem.groups,
                                  sem.use_squeeze_excite, sem.use_separable_conv)
        internals["aspp"] = aspp
        outs.append(seg)
    return (outs, internals) if return_internals else outs


# ----------------------------------------------------------------------------- deploy wrapper
def down_sample_input(images, target_size=(540, 960)):
    """DownSampleInput.call, engine/layers/misc.py:143-154."""
    x = np.asarray(images).astype(F32)                                   # :144 tf.cast(inputs, float32)
    _, ih, iw, _ = x.shape
    ratio = min(F32(target_size[0]) / F32(ih), F32(target_size[1]) / F32(iw))          # :150
    oh, ow = int(F32(ratio) * F32(ih)), int(F32(ratio) * F32(iw))                      # :151 cast -> int32
    return T.resize_bilinear_align_corners(x, oh, ow)                                  # :153


def trim_instances(roi_boxes, roi_masks, mold=True, max_batch_size=64):
    """TrimInstances.call, engine/layers/instance.py:258-277."""
    B = roi_boxes.shape[0]
    cls = roi_boxes[:, :, -2]
    idx = np.argwhere(cls != -1)                                          # :263 tf.where, row-major
    ci = cls[idx[:, 0], idx[:, 1]].astype(np.int64)                       # :265
    tm = np.transpose(roi_masks, (0, 1, 4, 2, 3))[idx[:, 0], idx[:, 1], ci]            # :268
    tb = roi_boxes[idx[:, 0], idx[:, 1]]                                  # :269
    if not mold:
        return tb, tm
    return (T.mold_batch(tb, idx[:, 0], B, None), T.mold_batch(tm, idx[:, 0], B, None))


def semantic_smoothing(x, kernel_size=10, weight=1.0):
    """SemanticSmoothing.call, engine/layers/semantic.py:270-285."""
    if kernel_size > 0:
        k = np.zeros((kernel_size, kernel_size, x.shape[-1]), F32)        # :273
        return T.dilation2d(T.erosion2d(x, k), k) * F32(weight)           # :275-284
    return x * F32(weight)


def up_sample_output(roi_box, roi_mask, semantic_output, target_hw):
    """UpSampleOutput.call, engine/layers/misc.py:169-196."""
    src = np.asarray(semantic_output.shape[1:3], F32)
    dst = np.asarray(target_hw, F32)
    ratio = dst / src                                                     # :176
    cx, cy, w, h, label, confs = [roi_box[..., i] for i in range(6)]
    box = np.stack([(cx * ratio[0]).astype(np.int32), (cy * ratio[1]).astype(np.int32),   # :179-182 (sic)
                    (w * ratio[0]).astype(np.int32), (h * ratio[1]).astype(np.int32),
                    label.astype(np.int32), (confs * F32(100)).astype(np.int32)], axis=-1)
    mask = (roi_mask > 0.5).astype(np.int32)                              # :188
    sem = T.resize_bilinear_align_corners(semantic_output.astype(F32), int(target_hw[0]), int(target_hw[1]))
    return box, mask, (sem > 0.5).astype(np.int32)                        # :193-194


def deploy_forward(config, weights, images, **kw):
    """The non-serving model of load_masklab_inference_model_from_h5, engine/retinamasklab.py:598-643:
    DownSampleInput -> inference model -> TrimInstances + per-class SemanticSmoothing + ResizeLike
    -> UpSampleOutput.  Returns int32 (detection, instance, semantic)."""
    images = np.asarray(images)
    down = down_sample_input(images, config.postprocess.resolution)       # :607
    _, _, box_pred, mask_pred, seg_pred = inference_forward(config, weights, down, **kw)   # :608
    det, inst = trim_instances(box_pred, mask_pred, mold=True)            # :614-615
    ks, ws = config.postprocess.smoothing_kernel_sizes, config.postprocess.smoothing_weights
    parts = np.split(seg_pred, len(ks), axis=-1)                          # :619-620
    post = np.concatenate([semantic_smoothing(t, k, w_) for t, k, w_ in zip(parts, ks, ws)], axis=-1)
    sem = T.resize_bilinear_align_corners(post, down.shape[1], down.shape[2])          # :628
    return up_sample_output(det, inst, sem, images.shape[1:3])            # :634-635


# ----------------------------------------------------------------------------- serving post-processing
def crop_and_pad_mask(images_hw, det_outs, ins_outs):
    """CropAndPadMask.call, engine/layers/misc.py:358-401.  det_outs int32 [B,n,6], ins_outs int32 [B,n,h,w]."""
    image_h, image_w = int(images_hw[0]), int(images_hw[1])
    B, n, _ = det_outs.shape
    threshold = 50 if det_outs[..., -1].max() > 50 else -100                       # :371-374
    out = np.zeros((B, n, image_h, image_w), F32)                                  # :396-398 scatter_nd into zeros
    for b, i in np.argwhere(det_outs[..., -1] >= threshold):                       # :375
        box = np.maximum(det_outs[b, i], 1).astype(F32)                            # :379, :382
        cx, cy, w, h = box[0], box[1], box[2], box[3]
        xmin = int(np.clip(np.int32(np.ceil(cx - w / F32(2))), 0, image_w))        # :384-391
        xmax = int(np.clip(np.int32(np.ceil(cx + w / F32(2))), 0, image_w))
        ymin = int(np.clip(np.int32(np.ceil(cy - h / F32(2))), 0, image_h))
        ymax = int(np.clip(np.int32(np.ceil(cy + h / F32(2))), 0, image_h))
        if ymax - ymin <= 0 or xmax - xmin <= 0:
            continue                                       # tf.image.resize to a zero size raises in the reference
        m = ins_outs[b, i].astype(F32)[None, :, :, None]
        r = T.resize_bilinear_align_corners(m, ymax - ymin, xmax - xmin)[0, :, :, 0]   # :393-395
        out[b, i, ymin:ymax, xmin:xmax] = r                                        # tf.pad (:396)
    return out


def crack_to_instance(crack):
    """CrackToInstance.call, engine/layers/misc.py:533-560.  crack int32 [B,H,W]."""
    idx = np.argwhere(crack)                                                       # :533
    if idx.size == 0:
        idx = np.zeros((1, 3), np.int64)                                           # :534-536
    ymin, xmin = idx.min(axis=0)[1:]
    ymax, xmax = idx.max(axis=0)[1:]
    height, width = np.int32(ymax - ymin), np.int32(xmax - xmin)
    cy = np.int32(ymin) + np.int32(height / 2)                                     # :542-543 (float division, then cast)
    cx = np.int32(xmin) + np.int32(width / 2)
    conf = np.int32(np.clip(100 * int(height) * int(width), 0, 100))               # :545
    row = np.asarray([cx, cy, width, height, 5, conf], np.int32)                   # `ones_like(cx) * 5` (:544)
    det = np.tile(row[None, None, :], (crack.shape[0], 1, 1))                      # :547-550
    return det, crack[:, None].astype(F32)                                         # :551


def _road_unit_length(image, default_road_size):
    """CalculateInstanceSize._calculate_road_size_by_vertical_per_batch, misc.py:663-679.  image [H,W] int32."""
    H = image.shape[0]
    idx = np.argwhere(image > 0)                                                   # :664
    ys, xs = idx[:, 0], idx[:, 1]
    n_seg = int(ys.max()) + 1 if len(ys) else 0
    x_mins = np.zeros(n_seg, np.int64)                                             # tf.segment_min / max: empty segment -> 0
    x_maxs = np.zeros(n_seg, np.int64)
    for y in np.unique(ys):
        row = xs[ys == y]
        x_mins[y], x_maxs[y] = row.min(), row.max()                                # :683-684
    y_pos = np.arange(n_seg, dtype=np.int64)
    keep = x_mins != x_maxs                                                        # :687-692
    left = np.stack([y_pos, x_mins], -1)[keep]
    right = np.stack([y_pos, x_maxs], -1)[keep]
    valid = F32(len(left))
    drop = int(np.clip(np.int32(valid * F32(0.15)), 1, 2 ** 31 - 1))               # :695-697
    left = left[drop:-drop].astype(F32)                                            # :699-702
    right = right[drop:-drop].astype(F32)

    def theta(pos):                                                                # :705-717
        xs_ = np.stack([pos[:, 0], np.ones_like(pos[:, 0])], axis=1).astype(F32)
        ys_ = pos[:, 1:2].astype(F32)
        x_mat = (xs_.T @ xs_).astype(F32)
        if np.linalg.det(x_mat.astype(F32)) > 0:
            return (np.linalg.inv(x_mat).astype(F32) @ (xs_.T @ ys_)).astype(F32)
        return np.zeros((2, 1), F32)

    lt, rt = theta(left), theta(right)
    y = np.arange(H, dtype=F32)
    width = np.clip((y * rt[0] + rt[1]) - (y * lt[0] + lt[1]), F32(1), np.inf).astype(F32)    # :672-677
    return (F32(default_road_size) / width).astype(F32)                            # :678


def calculate_instance_size(seg_outs, pad_ins_outs, default_road_size=3.25):
    """CalculateInstanceSize.call, engine/layers/misc.py:637-661 -> [B,n,3] (instance, horizontal, vertical)."""
    unit = np.stack([_road_unit_length(seg_outs[b, ..., 1], default_road_size) for b in range(seg_outs.shape[0])])   # :642-644
    m = pad_ins_outs.astype(F32)
    inst = ((unit ** 2)[:, None, :, None] * m).sum(axis=(2, 3))                    # :647-650
    vert = (unit[:, None, :] * (m > 0.5).any(axis=-1).astype(F32)).sum(axis=-1)    # :652-655
    horiz = (unit[:, None, :, None] * m).sum(axis=2).max(axis=-1)                  # :657-659
    return np.stack([inst, horiz, vert], axis=-1).astype(F32)


def include_my_road(seg_outs, crop_ins_outs, threshold=0.1):
    """IncludeMyRoad.call, engine/layers/misc.py:609-618."""
    my_road = seg_outs[..., 1].astype(F32)
    m = crop_ins_outs.astype(F32)
    inter = np.logical_and((my_road > 0.5)[:, None], m > 0.5).astype(F32).sum(axis=(2, 3))
    area = (m > 0.5).astype(F32).sum(axis=(2, 3))
    return ((inter / (area + F32(1e-5))) > F32(threshold)).astype(F32)


def summary_output(det_outs, seg_outs, crop_ins_outs, default_road_size=3.25):
    """SummaryOutput.call, engine/layers/misc.py:569-598 -> [B,n',11]."""
    crack_det, crack_seg = crack_to_instance(seg_outs[..., 2])                     # :575
    if np.all(crack_det[..., -1] > 0):                                             # :577-583
        det_outs = np.concatenate([det_outs, crack_det], axis=1)
        crop_ins_outs = np.concatenate([crop_ins_outs, crack_seg], axis=1)
    d = det_outs[..., :6].astype(F32)
    cx, cy, w, h, classes, conf = [d[..., i] for i in range(6)]                    # :585-586
    pixel_counts = crop_ins_outs.astype(F32).sum(axis=(2, 3))                      # :588-589
    sizes = calculate_instance_size(seg_outs, crop_ins_outs, default_road_size)    # :591-593
    inc = include_my_road(seg_outs, crop_ins_outs)                                 # :595
    return np.stack([classes, cx, cy, w, h, conf, pixel_counts, sizes[..., 0], sizes[..., 1], sizes[..., 2], inc],
                    axis=-1).astype(F32)                                           # :597-598


def serving_forward(config, weights, images, **kw):
    """The 'summarize' output of load_serving_model_from_h5, road_project/setup/serving.py:27-48."""
    images = np.asarray(images)
    det, inst, sem = deploy_forward(config, weights, images, **kw)                 # :24-29
    masks = crop_and_pad_mask(images.shape[1:3], det, inst)                        # :30
    return summary_output(det, sem, masks, config.postprocess.default_road_size)  # :47-48
