"""CPU tests of the oracle: the anchor table against the REFERENCE's own output (the only pinned
fixture), and every restated TF op against an independent formulation (brute-force scalar loops
or torch-CPU).  See oracle/tfops.py header: parity vs TensorFlow itself is unpinned."""
import math
import os

import numpy as np
import pytest

from oracle import masklab as O
from oracle import tfops as T

RNG = np.random.default_rng(3)


def rnd(*shape):
    return RNG.normal(size=shape).astype(np.float32)


# ------------------------------------------------------------------ reference-pinned: anchors
@pytest.mark.parametrize("case", ["default", "three_level", "two_scale", "unsorted_strides"])
def test_prior_table_matches_reference_output(golden_dir, case):
    g = np.load(os.path.join(golden_dir, "prior_tables.npz"))
    table = O.prior_table(g[case + "/strides"].tolist(), g[case + "/sizes"].tolist(),
                          g[case + "/scales"].tolist(), g[case + "/ratios"].tolist())
    np.testing.assert_array_equal(table, g[case + "/table"])
    assert len(g[case + "/scales"]) * len(g[case + "/ratios"]) == int(g[case + "/len"])
    # the product's PriorBoxes is a separate implementation: pin it too
    from masklab_hip import PriorBoxes
    pb = PriorBoxes(g[case + "/strides"], g[case + "/sizes"], g[case + "/scales"], g[case + "/ratios"])
    np.testing.assert_array_equal(pb.table, g[case + "/table"])
    assert len(pb) == int(g[case + "/len"])
    np.testing.assert_array_equal(pb.boxes.values, g[case + "/table"])


def test_prior_boxes_layout_and_count():
    table = O.prior_table([8, 16, 32, 64, 128], [32, 64, 128, 256, 512],
                          [2 ** 0, 2 ** (1 / 3), 2 ** (2 / 3)], [1 / 3, 1 / 2, 1, 2, 3])
    a = O.prior_boxes(table, 1024, 1024)
    assert a.shape == (327360, 4) and a.dtype == np.int32
    assert O.prior_boxes(table, 512, 512).shape[0] == 81840
    # order inside a level: (y, x, anchor); first cell of P3 carries the 15 table rows at (4, 4)
    np.testing.assert_array_equal(a[:15, :2], np.full((15, 2), 4))
    np.testing.assert_array_equal(a[:15, 2:], table[:15, 1:])
    np.testing.assert_array_equal(a[15, :2], [12, 4])                # next cell moves along x
    from masklab_hip import PriorBoxes
    pb = PriorBoxes([8, 16, 32, 64, 128], [32, 64, 128, 256, 512], [2 ** 0, 2 ** (1 / 3), 2 ** (2 / 3)],
                    [1 / 3, 1 / 2, 1, 2, 3])
    for hw in [(128, 128), (128, 384), (200, 136)]:
        np.testing.assert_array_equal(pb.anchors(*hw), O.prior_boxes(table, *hw))


# ------------------------------------------------------------------ conv family vs torch CPU
torch = pytest.importorskip("torch")
F = torch.nn.functional


def _nchw(x):
    return torch.from_numpy(np.ascontiguousarray(x.transpose(0, 3, 1, 2))).double()


@pytest.mark.parametrize("k,stride,padding,dil", [(3, 1, "same", 1), (3, 2, "same", 1), (1, 2, "valid", 1),
                                                  (7, 2, ((3, 3), (3, 3)), 1), (3, 2, ((0, 1), (0, 1)), 1),
                                                  (3, 1, "same", 6)])
@pytest.mark.parametrize("hw", [(12, 12), (11, 13)])
def test_conv2d_vs_torch(k, stride, padding, dil, hw):
    x, w, b = rnd(2, hw[0], hw[1], 5), rnd(k, k, 5, 7), rnd(7)
    got = T.conv2d(x.astype(np.float64), w, b, stride, padding, dil)
    Ho, Wo, pt, pb, pl, pr = T._resolve_padding(x, k, k, stride, dil, padding)
    xp = F.pad(_nchw(x), (pl, pr, pt, pb))
    ref = F.conv2d(xp, torch.from_numpy(w.transpose(3, 2, 0, 1).copy()).double(), torch.from_numpy(b).double(),
                   stride=stride, dilation=dil).numpy().transpose(0, 2, 3, 1)
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, atol=1e-10)


def test_same_padding_is_asymmetric_for_stride_2():
    assert T.same_pads(16, 3, 2) == (8, 0, 1)        # even: extra pixel after
    assert T.same_pads(15, 3, 2) == (8, 1, 1)
    assert T.same_pads(32, 3, 1, 6) == (32, 6, 6)    # dilated: pad = rate


def test_depthwise_and_transpose_vs_torch():
    x, w = rnd(2, 9, 10, 6), rnd(3, 3, 6, 2)
    got = T.depthwise_conv2d(x.astype(np.float64), w, 1, "same", 2)
    wt = torch.from_numpy(w.transpose(2, 3, 0, 1).reshape(12, 1, 3, 3).copy()).double()   # out = cin*mult + m
    ref = F.conv2d(F.pad(_nchw(x), (2, 2, 2, 2)), wt, dilation=2, groups=6).numpy().transpose(0, 2, 3, 1)
    np.testing.assert_allclose(got, ref, atol=1e-10)
    xt, wt2, b = rnd(2, 5, 4, 6), rnd(2, 2, 3, 6), rnd(3)
    got = T.conv2d_transpose_2x2_s2(xt.astype(np.float64), wt2, b)
    ref = F.conv_transpose2d(_nchw(xt), torch.from_numpy(wt2.transpose(3, 2, 0, 1).copy()).double(),
                             torch.from_numpy(b).double(), stride=2).numpy().transpose(0, 2, 3, 1)
    np.testing.assert_allclose(got, ref, atol=1e-10)


def test_max_pool_vs_torch():
    x = np.abs(rnd(2, 11, 12, 4))
    got = T.max_pool(np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0))), 3, 2)
    ref = F.max_pool2d(_nchw(x), 3, 2, padding=1).numpy().transpose(0, 2, 3, 1)
    np.testing.assert_allclose(got, ref)


# ------------------------------------------------------------------ GroupNormalization quirk
@pytest.mark.parametrize("shape,G", [((2, 4, 4, 32), 16), ((1, 14, 14, 128), 16), ((2, 2, 2, 128), 32),
                                     ((1, 6, 5, 12), 3)])
def test_group_norm_literal_equals_flat_chunks(shape, G):
    x = rnd(*shape)
    gamma, beta = RNG.uniform(0.5, 1.5, shape[-1]).astype(np.float32), rnd(shape[-1])
    np.testing.assert_allclose(T.group_norm(x.astype(np.float64), gamma, beta, G),
                               T.group_norm_flat(x, gamma, beta, G), atol=1e-9)


def test_group_norm_is_not_channel_grouping():
    """with axis=-1 on NHWC the reference normalises contiguous chunks of the flat HWC
    vector -- NOT channel groups.  A textbook GroupNorm gives a different answer."""
    x = rnd(1, 8, 8, 32)
    ones, zeros = np.ones(32, np.float32), np.zeros(32, np.float32)
    ours = T.group_norm(x.astype(np.float64), ones, zeros, 16)
    textbook = F.group_norm(_nchw(x), 16, eps=1e-5).numpy().transpose(0, 2, 3, 1)
    assert np.max(np.abs(ours - textbook)) > 0.1
    # chunk statistics: every contiguous chunk of H*W*C/G elements has mean 0 / var ~1
    chunks = ours.reshape(1, 16, -1)
    np.testing.assert_allclose(chunks.mean(-1), 0, atol=1e-9)
    np.testing.assert_allclose(chunks.var(-1), 1, atol=1e-3)


def test_group_norm_errors_match_reference_messages():
    with pytest.raises(ValueError, match="cannot be more than the number of channels"):
        T.group_norm(rnd(1, 2, 2, 8), np.ones(8), np.zeros(8), 16)
    with pytest.raises(ValueError, match="must be a multiple of the number of channels"):
        T.group_norm(rnd(1, 2, 2, 12), np.ones(12), np.zeros(12), 5)


# ------------------------------------------------------------------ resampling vs scalar loops
def test_resize_bilinear_align_corners_vs_loops():
    x = rnd(1, 4, 5, 2).astype(np.float64)
    oh, ow = 9, 7
    got = T.resize_bilinear_align_corners(x, oh, ow)
    ref = np.zeros((1, oh, ow, 2))
    for i in range(oh):
        sy = np.float32(i) * np.float32((4 - 1) / float(oh - 1))
        y0, y1, ty = int(math.floor(sy)), min(int(math.ceil(sy)), 3), sy - math.floor(sy)
        for j in range(ow):
            sx = np.float32(j) * np.float32((5 - 1) / float(ow - 1))
            x0, x1, tx = int(math.floor(sx)), min(int(math.ceil(sx)), 4), sx - math.floor(sx)
            top = x[0, y0, x0] + (x[0, y0, x1] - x[0, y0, x0]) * tx
            bot = x[0, y1, x0] + (x[0, y1, x1] - x[0, y1, x0]) * tx
            ref[0, i, j] = top + (bot - top) * ty
    np.testing.assert_allclose(got, ref, atol=1e-6)
    np.testing.assert_allclose(got[0, 0, 0], x[0, 0, 0]); np.testing.assert_allclose(got[0, -1, -1], x[0, -1, -1])
    one = rnd(2, 1, 1, 3).astype(np.float64)                        # 1x1 source = broadcast (ASPP pool)
    np.testing.assert_allclose(T.resize_bilinear_align_corners(one, 4, 6), np.broadcast_to(one, (2, 4, 6, 3)))
    same = T.resize_bilinear_align_corners(x, 4, 5)
    np.testing.assert_allclose(same, x)


def test_crop_and_resize_vs_loops_and_extrapolation():
    img = rnd(2, 6, 7, 3)
    boxes = np.asarray([[0.1, 0.2, 0.8, 0.9], [-0.2, 0.0, 0.5, 1.3], [0.0, 0.0, 1.0, 1.0], [0.9, 0.9, 0.1, 0.1]],
                       np.float32)
    ind = np.asarray([0, 1, 1, 0])
    got = T.crop_and_resize(img, boxes, ind, (4, 5))
    H, W = 6, 7
    for r, (y1, x1, y2, x2) in enumerate(boxes.astype(np.float64)):
        for i in range(4):
            iy = y1 * (H - 1) + i * (y2 - y1) * (H - 1) / 3
            for j in range(5):
                ix = x1 * (W - 1) + j * (x2 - x1) * (W - 1) / 4
                if iy < 0 or iy > H - 1 or ix < 0 or ix > W - 1:
                    want = np.zeros(3)
                else:
                    t, b_, l, rr = math.floor(iy), math.ceil(iy), math.floor(ix), math.ceil(ix)
                    im = img[ind[r]].astype(np.float64)
                    top = im[t, l] + (im[t, rr] - im[t, l]) * (ix - l)
                    bot = im[b_, l] + (im[b_, rr] - im[b_, l]) * (ix - l)
                    want = top + (bot - top) * (iy - t)
                np.testing.assert_allclose(got[r, i, j], want, atol=2e-5)
    # full-image box reproduces the corners exactly (normalised by H-1 / W-1 inside the op)
    np.testing.assert_allclose(got[2, 0, 0], img[1, 0, 0]); np.testing.assert_allclose(got[2, -1, -1], img[1, -1, -1])


# ------------------------------------------------------------------ NMS / mold
def _iou_plain(a, b):
    ya1, xa1, ya2, xa2 = min(a[0], a[2]), min(a[1], a[3]), max(a[0], a[2]), max(a[1], a[3])
    yb1, xb1, yb2, xb2 = min(b[0], b[2]), min(b[1], b[3]), max(b[0], b[2]), max(b[1], b[3])
    aa, ab = (ya2 - ya1) * (xa2 - xa1), (yb2 - yb1) * (xb2 - xb1)
    if aa <= 0 or ab <= 0:
        return 0.0
    inter = max(min(ya2, yb2) - max(ya1, yb1), 0) * max(min(xa2, xb2) - max(xa1, xb1), 0)
    return inter / (aa + ab - inter)


def test_nms_vs_brute_force_and_known_answer():
    rng = np.random.default_rng(9)
    c = rng.uniform(0, 50, (60, 2))
    s = rng.uniform(5, 25, (60, 2))
    boxes = np.concatenate([c - s / 2, c + s / 2], 1).astype(np.float32)
    scores = rng.permutation(60).astype(np.float32) / 60
    got = T.non_max_suppression(boxes, scores, 10, 0.3)
    keep = []
    for i in np.argsort(-scores):
        if len(keep) == 10:
            break
        if all(_iou_plain(boxes[i].astype(np.float64), boxes[k].astype(np.float64)) <= 0.3 for k in keep):
            keep.append(i)
    np.testing.assert_array_equal(got, keep)
    # known answer: two heavy overlaps + one far box; flipped corners tolerated; strict '>' at the threshold
    b = np.asarray([[0, 0, 10, 10], [1, 1, 11, 11], [50, 50, 60, 60], [10, 10, 0, 0]], np.float32)
    sc = np.asarray([0.9, 0.8, 0.7, 0.6], np.float32)
    np.testing.assert_array_equal(T.non_max_suppression(b, sc, 10, 0.5), [0, 2])
    iou01 = _iou_plain(b[0], b[1])
    np.testing.assert_array_equal(T.non_max_suppression(b[:2], sc[:2], 10, np.float32(iou01)), [0, 1])
    np.testing.assert_array_equal(T.non_max_suppression(b, sc, 1, 0.5), [0])
    # equal scores: lower index first (documented tie rule)
    np.testing.assert_array_equal(T.non_max_suppression(b[[2, 0]], np.asarray([0.5, 0.5], np.float32), 5, 0.5), [0, 1])
    # degenerate (zero-area) boxes never suppress each other
    z = np.asarray([[5, 5, 5, 9], [5, 5, 5, 9]], np.float32)
    np.testing.assert_array_equal(T.non_max_suppression(z, np.asarray([0.9, 0.8], np.float32), 5, 0.1), [0, 1])


def test_mold_batch_edge_cases():
    rows = np.arange(12, dtype=np.float32).reshape(4, 3)
    out = T.mold_batch(rows, np.asarray([2, 0, 2, 2]), 3)
    assert out.shape == (3, 3, 3)
    np.testing.assert_array_equal(out[0, 0], rows[1]); np.testing.assert_array_equal(out[2, :3], rows[[0, 2, 3]])
    assert np.all(out[1] == -1) and np.all(out[0, 1:] == -1)
    empty = T.mold_batch(np.zeros((0, 6), np.float32), np.zeros((0,), np.int64), 2)
    assert empty.shape == (2, 1, 6) and np.all(empty == -1)           # reference misc.py:236 max(1, .)
    with pytest.raises(ValueError):
        T.mold_batch(rows, np.zeros(4, np.int64), 33)                 # 32-slot dynamic_partition


def test_detection_proposal_hand_made_case():
    """two classes, overlapping boxes: per-class NMS, then cross-class NMS, score-descending output"""
    A, C = 6, 2
    boxes = np.asarray([[[10, 10, 10, 10], [11, 10, 10, 10], [40, 40, 10, 10],
                         [10, 10, 10, 10], [70, 70, 8, 8], [41, 40, 10, 10]]], np.float32)   # (cx,cy,w,h)
    cls = np.zeros((1, A, C), np.float32)
    cls[0, 0, 0] = 0.9     # kept (best of its cluster, class 0)
    cls[0, 1, 0] = 0.8     # suppressed by anchor 0 in the per-class NMS (IoU 0.82 > 0.4)
    cls[0, 2, 0] = 0.7     # kept
    cls[0, 3, 1] = 0.95    # class 1, identical box to anchor 0 -> wins the cross-class NMS (IoU 1 > 0.6)
    cls[0, 4, 1] = 0.6     # kept
    cls[0, 5, 1] = 0.55    # class 1, IoU with anchor 2 = 0.82 > 0.6 -> removed by cross-class NMS
    out, kept = O.detection_proposal(cls, boxes, 0.5, 0.4, 0.6, 100)
    np.testing.assert_array_equal(kept, [[0, 3, 1], [0, 2, 0], [0, 4, 1]])
    assert out.shape == (1, 3, 6)
    np.testing.assert_allclose(out[0, :, 5], [0.95, 0.7, 0.6])
    np.testing.assert_array_equal(out[0, :, 4], [1, 0, 1])
    none, k0 = O.detection_proposal(cls * 0, boxes, 0.5, 0.4, 0.6, 100)
    assert none.shape == (1, 1, 6) and np.all(none == -1) and len(k0) == 0


def test_mask_distribute_levels_and_padding():
    p = np.full((1, 5, 6), -1.0, np.float32)
    p[0, 0, :4] = [50, 50, 36, 36]       # size 36 -> k 0
    p[0, 1, :4] = [50, 50, 72, 72]       # size 72 -> k 1
    p[0, 2, :4] = [50, 50, 300, 300]     # clipped to max_k
    p[0, 3, :4] = [50, 50, 10, 10]       # below base -> clipped to 0
    d = O.mask_distribute(p, 2, 36)
    np.testing.assert_array_equal(d[0, :, 0], [0, 1, 2, 0, -1])
    np.testing.assert_array_equal(d[..., 1:], p)


def test_grouped_conv_literal_equals_fast():
    x, k = rnd(1, 6, 7, 32), rnd(3, 3, 32, 4)
    for stride in (1, 2):
        np.testing.assert_allclose(O.grouped_conv_literal(x.astype(np.float64), k, 8, 4, stride),
                                   O.grouped_conv_fast(x.astype(np.float64), k, 8, 4, stride), atol=1e-10)


def test_preprocess_modes():
    img = RNG.integers(0, 256, (1, 2, 2, 3)).astype(np.float32)
    np.testing.assert_allclose(O.backbone_preprocess(img, rgb=True, mean_shift=True, normalize=2),
                               (img - np.asarray([123.68, 116.779, 103.939], np.float32)) / 127.5, rtol=1e-6)
    np.testing.assert_allclose(O.backbone_preprocess(img, rgb=False, mean_shift=False, normalize=2),
                               img[..., ::-1] / 127.5 - 1, rtol=1e-6)


# ----------------------------------------------------------------------------- deploy wrapper
@pytest.mark.parametrize("k", [1, 2, 3, 4, 10, 11])
def test_morphology_against_scipy(k):
    """Independent restatement: scipy.ndimage min/max filters with the window shifted the way TF's SAME
    padding places it (rows y-(k-1)//2 .. y-(k-1)//2+k-1) and +-inf outside the map."""
    ndi = pytest.importorskip("scipy.ndimage")
    x = np.random.default_rng(k).normal(size=(2, 13, 17, 3)).astype(np.float32)
    z = np.zeros((k, k, 3), np.float32)
    org = 0 if k % 2 else -1
    want_d = ndi.maximum_filter(x, size=(1, k, k, 1), mode="constant", cval=-np.inf, origin=(0, org, org, 0))
    want_e = ndi.minimum_filter(x, size=(1, k, k, 1), mode="constant", cval=np.inf, origin=(0, org, org, 0))
    np.testing.assert_array_equal(T.dilation2d(x, z), want_d)
    np.testing.assert_array_equal(T.erosion2d(x, z), want_e)


def test_dilation_with_nonzero_structuring_element():
    """hand-checked 1-D case: in = [1, 5, 2], element [0, 10, 0] -> max(in[x-1], in[x]+10, in[x+1])"""
    x = np.array([1, 5, 2], np.float32).reshape(1, 1, 3, 1)
    k = np.array([0, 10, 0], np.float32).reshape(1, 3, 1)
    np.testing.assert_array_equal(T.dilation2d(x, k).ravel(), [11, 15, 12])
    # erosion by duality with the element reversed: min(in[x-1]-0, in[x]-10, in[x+1]-0)
    np.testing.assert_array_equal(T.erosion2d(x, k).ravel(), [-9, -5, -8])


def test_semantic_smoothing_is_an_opening():
    x = np.random.default_rng(0).random((1, 20, 20, 2)).astype(np.float32)
    y = O.semantic_smoothing(x, 3, 1.0)
    assert np.all(y <= x)                                           # an opening never exceeds its input
    np.testing.assert_array_equal(O.semantic_smoothing(y, 3, 1.0), y)     # and is idempotent
    np.testing.assert_array_equal(O.semantic_smoothing(x, 0, 2.0), x * np.float32(2.0))


def test_down_sample_input_size_rule():
    for (h, w), want in (((1080, 1920), (540, 960)), ((720, 960), (540, 720)), ((400, 640), (540, 864)),
                         ((1000, 3000), (320, 960))):
        out = O.down_sample_input(np.zeros((1, h, w, 3), np.uint8), (540, 960))
        assert out.shape[1:3] == want and out.dtype == np.float32


def test_trim_instances_orders_and_pads():
    boxes = np.full((2, 4, 6), -1, np.float32)
    boxes[0, 0] = [10, 10, 4, 4, 2, 0.9]
    boxes[0, 2] = [20, 20, 4, 4, 0, 0.8]                            # a hole at row 1: order must be kept
    masks = np.arange(2 * 4 * 2 * 2 * 3, dtype=np.float32).reshape(2, 4, 2, 2, 3)
    b, m = O.trim_instances(boxes, masks)
    assert b.shape == (2, 2, 6) and m.shape == (2, 2, 2, 2)
    np.testing.assert_array_equal(b[0], boxes[0, [0, 2]])
    np.testing.assert_array_equal(m[0, 0], masks[0, 0, :, :, 2])
    np.testing.assert_array_equal(m[0, 1], masks[0, 2, :, :, 0])
    assert np.all(b[1] == -1) and np.all(m[1] == -1)


def test_up_sample_output_axis_quirk():
    """cx and w are scaled by the HEIGHT ratio, cy and h by the width ratio (misc.py:180-183)."""
    box = np.array([[[10, 10, 4, 4, 1, 0.876], [-1, -1, -1, -1, -1, -1]]], np.float32)
    mask = np.zeros((1, 2, 2, 2), np.float32)
    sem = np.zeros((1, 10, 20, 1), np.float32)
    b, m, s = O.up_sample_output(box, mask, sem, (30, 40))           # ratios: height 3, width 2
    np.testing.assert_array_equal(b[0, 0], [30, 20, 12, 8, 1, 87])
    np.testing.assert_array_equal(b[0, 1], [-3, -2, -3, -2, -1, -100])
    assert s.shape == (1, 30, 40, 1) and s.dtype == np.int32


# ----------------------------------------------------------------------------- detection metric
def test_detection_iou_metric_restatement():
    from oracle import metrics as M
    gt = np.full((2, 3, 6), -1, np.float32)
    gt[0, 0] = [50, 50, 20, 20, 1, 1.0]
    gt[0, 1] = [150, 50, 40, 20, 2, 1.0]
    gt[1, 0] = [30, 30, 10, 10, 0, 1.0]
    pred = np.full((2, 4, 6), -1, np.float32)
    pred[0, 0] = [51, 50, 20, 20, 1, 0.9]           # IoU 19*20/(800-380) = 0.905 with gt 0
    pred[0, 1] = [150, 80, 40, 20, 2, 0.8]          # no overlap with gt 1 (30 px lower)
    pred[0, 2] = [150, 55, 40, 20, 2, 0.7]          # IoU 15*40/(1600-600) = 0.6 with gt 1
    pred[1, 0] = [30, 30, 10, 10, 0, 0.9]
    iou = M.calculate_iou(pred[0], gt[0])
    np.testing.assert_allclose(iou[0, 0], 380.0 / 420.0, rtol=1e-5)
    np.testing.assert_allclose(iou[2, 1], 600.0 / 1000.0, rtol=1e-5)
    assert iou[1, 1] == 0
    p, r, f = M.detection_iou_metric(pred, gt)
    np.testing.assert_allclose(p, [2 / 3, 1.0], rtol=1e-6)
    np.testing.assert_allclose(r, [1.0, 1.0], rtol=1e-6)
    np.testing.assert_allclose(f, [0.8, 1.0], rtol=1e-6)
    # identical inputs score 1 (up to the metric's epsilon), an image with nothing on either side scores 0
    p, r, f = M.detection_iou_metric(gt, gt)
    np.testing.assert_allclose(p, 1.0, rtol=1e-6)
    empty = np.full((1, 2, 6), -1, np.float32)
    p, r, f = M.detection_iou_metric(empty, empty)
    assert p[0] == 0 and r[0] == 0 and f[0] == 0


# ----------------------------------------------------------------------------- serving post-processing
def test_crop_and_pad_mask_hand_case():
    det = np.array([[[6, 5, 4, 2, 1, 90], [3, 3, 2, 2, 0, 40]]], np.int32)        # second row below the conf-50 cut
    ins = np.zeros((1, 2, 2, 2), np.int32)
    ins[0, 0] = [[1, 0], [0, 1]]
    ins[0, 1] = 1
    out = O.crop_and_pad_mask((8, 10), det, ins)
    assert out.shape == (1, 2, 8, 10) and out.dtype == np.float32
    # box 0: x in [ceil(6-2), ceil(6+2)) = [4, 8), y in [ceil(5-1), ceil(5+1)) = [4, 6): the 2x2 mask resized to 2x4
    want = np.zeros((8, 10), np.float32)
    want[4, 4:8] = [1, 2 / 3, 1 / 3, 0]
    want[5, 4:8] = [0, 1 / 3, 2 / 3, 1]
    np.testing.assert_allclose(out[0, 0], want, atol=1e-6)
    assert out[0, 1].max() == 0                                                  # not selected
    det[0, 0, 5] = 45                                                            # nothing above 50 -> every row is kept
    out = O.crop_and_pad_mask((8, 10), det, ins)
    assert out[0, 1, 2:4, 2:4].min() == 1 and out[0, 1].sum() == 4


def test_crack_to_instance_and_summary_columns():
    seg = np.zeros((2, 12, 16, 3), np.int32)
    seg[0, 3:7, 4:10, 2] = 1                                                     # crack pixels: y 3..6, x 4..9
    seg[1, 8, 12, 2] = 1                                                         # the bounding box spans the BATCH
    det, cseg = O.crack_to_instance(seg[..., 2])
    # ymin 3, ymax 8, xmin 4, xmax 12 -> h 5, w 8, cy 3 + 2, cx 4 + 4, conf clip(100*5*8) = 100
    np.testing.assert_array_equal(det[:, 0], [[8, 5, 8, 5, 5, 100]] * 2)
    assert cseg.shape == (2, 1, 12, 16) and cseg.sum() == 25
    det0 = np.array([[[8, 6, 4, 4, 2, 80]], [[8, 6, 4, 4, 2, 80]]], np.int32)
    masks = np.zeros((2, 1, 12, 16), np.float32)
    masks[:, 0, 4:8, 6:10] = 1
    seg[:, 2:11, 5:11, 1] = 1                                                    # a straight road, 6 px wide: x 5..10
    out = O.summary_output(det0, seg, masks, default_road_size=3.0)
    assert out.shape == (2, 2, 11)
    np.testing.assert_array_equal(out[0, 0, :7], [2, 8, 6, 4, 4, 80, 16])        # class, cx, cy, w, h, conf, pixels
    # edges: left x = 5, right x = 10 for every kept row -> width 5 -> unit 0.6; 16 px * 0.36, 4 rows * 0.6, 4 px * 0.6
    np.testing.assert_allclose(out[0, 0, 7:10], [16 * 0.36, 4 * 0.6, 4 * 0.6], rtol=1e-4)
    assert out[0, 0, 10] == 1 and out[0, 1, 0] == 5                              # on the road; crack pseudo instance appended
    none = O.summary_output(det0, np.zeros_like(seg), masks)
    assert none.shape == (2, 1, 11)                                              # no crack pixels: nothing appended
    np.testing.assert_allclose(none[0, 0, 7:10], [16 * 3.25 ** 2, 4 * 3.25, 4 * 3.25], rtol=1e-6)   # no road: width clipped to 1


def test_road_regression_matches_float64_least_squares():
    """the float32 normal-equation solve of the reference against numpy's float64 lstsq on a clean trapezoid"""
    H, W = 200, 300
    img = np.zeros((H, W), np.int32)
    for y in range(40, 190):
        img[y, int(100 - 0.3 * (y - 40)):int(180 + 0.4 * (y - 40))] = 1
    unit = O._road_unit_length(img, 3.25)
    ys = np.arange(40, 190)[22:-22].astype(np.float64)                          # 15 % of 150 rows dropped at both ends
    lx = np.array([np.nonzero(img[int(y)])[0].min() for y in ys], np.float64)
    rx = np.array([np.nonzero(img[int(y)])[0].max() for y in ys], np.float64)
    A = np.stack([ys, np.ones_like(ys)], 1)
    lt, rt = np.linalg.lstsq(A, lx, rcond=None)[0], np.linalg.lstsq(A, rx, rcond=None)[0]
    y = np.arange(H)
    want = 3.25 / np.clip((y * rt[0] + rt[1]) - (y * lt[0] + lt[1]), 1, np.inf)
    np.testing.assert_allclose(unit, want, rtol=2e-3)


# ------------------------------------------------------------------ independent implementations
# The oracle cannot be pinned to TensorFlow here (no TF, no reference fixtures), so every restated op is
# also checked against an implementation that shares NO code with oracle/: torch-CPU library kernels and a second,
# loop-form DetectionProposal written straight from the reference text.
@pytest.mark.parametrize("shape,out", [((2, 5, 7, 4), (11, 13)), ((1, 32, 32, 8), (128, 128)), ((1, 1, 1, 6), (9, 4)),
                                       ((2, 9, 6, 3), (9, 6)), ((1, 16, 16, 4), (5, 3))])
def test_resize_bilinear_vs_torch_interpolate_align_corners(shape, out):
    x = rnd(*shape)
    ref = F.interpolate(_nchw(x.astype(np.float64)), size=out, mode="bilinear", align_corners=True)
    # the oracle computes the source coordinate o*(in-1)/(out-1) in fp32 like TF's kernel does: a position error of
    # <= 1 ulp(in) ~ 1e-5 moves a value by <= 1e-5 * |neighbour difference|
    np.testing.assert_allclose(T.resize_bilinear_align_corners(x, *out), ref.numpy().transpose(0, 2, 3, 1), atol=4e-5)


@pytest.mark.parametrize("shape,G", [((2, 4, 4, 32), 16), ((1, 14, 14, 128), 16), ((3, 8, 8, 128), 32), ((1, 6, 5, 12), 3)])
def test_group_norm_vs_torch_group_norm_on_the_5d_view(shape, G):
    """normalization.py:123-143 reshapes NHWC row-major to [N,G,H,W,C/G] and normalises over the last three axes:
    on that 5-D view it is torch's group_norm with G 'channels groups' of one channel each; gamma / beta are
    broadcast as [1,G,1,1,C/G] (:151-156)."""
    x = rnd(*shape)
    N, H, W, C = shape
    gamma, beta = RNG.uniform(0.5, 1.5, C).astype(np.float32), rnd(C)
    v = torch.from_numpy(x.astype(np.float64)).reshape(N, G, H * W * C // G)          # [N, G channels, L]
    y = F.group_norm(v, G, eps=1e-5).reshape(N, G, H, W, C // G)
    y = y * torch.from_numpy(gamma.astype(np.float64)).reshape(1, G, 1, 1, C // G) + \
        torch.from_numpy(beta.astype(np.float64)).reshape(1, G, 1, 1, C // G)
    np.testing.assert_allclose(T.group_norm(x.astype(np.float64), gamma, beta, G), y.reshape(N, H, W, C).numpy(),
                               atol=1e-9)


@pytest.mark.parametrize("stride,c,groups", [(1, 4, 32), (2, 4, 32), (1, 8, 32), (2, 16, 32)])
def test_grouped_conv_vs_torch_conv2d_groups(stride, c, groups):
    """DepthwiseConv2D(depth_multiplier=c) + reshape + reduce_sum (ResNext.py:212-219) IS a grouped
    convolution with Wg[kh,kw,i,g*c+m] = K[kh,kw,g*c+i,m]; checked against torch's own grouped conv."""
    filters = groups * c
    x, k = rnd(2, 9, 10, filters), rnd(3, 3, filters, c)
    lit = O.grouped_conv_literal(x.astype(np.float64), k.astype(np.float64), groups, c, stride)
    # torch weight [out, in/groups, kh, kw]: out channel g*c+m, in-group channel i  <-  K[:, :, g*c+i, m]
    wt = np.zeros((filters, c, 3, 3))
    for g in range(groups):
        for m in range(c):
            for i in range(c):
                wt[g * c + m, i] = k[:, :, g * c + i, m]
    ref = F.conv2d(_nchw(x.astype(np.float64)), torch.from_numpy(wt), stride=stride, padding=1, groups=groups)
    np.testing.assert_allclose(lit, ref.numpy().transpose(0, 2, 3, 1), atol=1e-10)
    np.testing.assert_allclose(O.grouped_conv_fast(x.astype(np.float64), k, groups, c, stride), lit, atol=1e-10)


def _detection_proposal_loops(cls_pred, boxes, min_conf, nms_iou, post_iou, max_out):
    """Second restatement of DetectionProposal.call, written line by line from reference detection.py:482-567 with
    plain Python loops and its own IoU / greedy NMS (shares nothing with oracle.masklab.detection_proposal or
    oracle.tfops.non_max_suppression)."""
    f32 = np.float32

    def iou(a, b):                                           # tf non_max_suppression: y1,x1,y2,x2, area<=0 -> 0
        ay1, ax1, ay2, ax2 = min(a[0], a[2]), min(a[1], a[3]), max(a[0], a[2]), max(a[1], a[3])
        by1, bx1, by2, bx2 = min(b[0], b[2]), min(b[1], b[3]), max(b[0], b[2]), max(b[1], b[3])
        aa, ab = f32(ay2 - ay1) * f32(ax2 - ax1), f32(by2 - by1) * f32(bx2 - bx1)
        if aa <= 0 or ab <= 0:
            return f32(0)
        ih = max(f32(min(ay2, by2) - max(ay1, by1)), f32(0))
        iw = max(f32(min(ax2, bx2) - max(ax1, bx1)), f32(0))
        inter = f32(ih * iw)
        return f32(inter / f32(f32(aa + ab) - inter))

    def nms(cands, thr):                                     # cands: list of (score, tie index, corner box, payload)
        order = sorted(range(len(cands)), key=lambda i: (-cands[i][0], cands[i][1]))
        keep = []
        for i in order:
            if len(keep) >= max_out:
                break
            if all(not (iou(cands[i][2], cands[j][2]) > f32(thr)) for j in keep):
                keep.append(i)
        return [cands[i] for i in keep]

    B, A, C = cls_pred.shape
    corner = np.empty_like(boxes)                            # NormalizeBoxes without shape (:488, :362-374)
    corner[..., 0] = boxes[..., 1] - boxes[..., 3] / f32(2)
    corner[..., 1] = boxes[..., 0] - boxes[..., 2] / f32(2)
    corner[..., 2] = boxes[..., 1] + boxes[..., 3] / f32(2)
    corner[..., 3] = boxes[..., 0] + boxes[..., 2] / f32(2)
    per_id, seen = {}, []                                    # :491-520: tf.where row-major, tf.unique first occurrence
    for b in range(B):
        for a in range(A):
            for c in range(C):
                if cls_pred[b, a, c] >= f32(min_conf):
                    key = b * (C + 1) + c
                    if key not in per_id:
                        per_id[key] = []
                        seen.append(key)
                    per_id[key].append((cls_pred[b, a, c], a, corner[b, a], (b, a, c)))
    stage1 = []
    for key in seen:                                         # :522 map_fn, :507-514
        stage1 += nms(per_id[key], nms_iou)
    final = []
    for b in range(B):                                       # :531-555
        cand = [(s, pos, cb, pl) for pos, (s, _, cb, pl) in enumerate(stage1) if pl[0] == b]
        final += [pl for (_, _, _, pl) in nms(cand, post_iou)]
    return np.array(final, np.int64).reshape(-1, 3)


@pytest.mark.parametrize("seed,frac", [(0, 0.02), (1, 0.10), (2, 0.30)])
def test_detection_proposal_vs_loop_form(seed, frac):
    rng = np.random.default_rng(seed)
    B, C = 2, 5
    table = O.prior_table([8, 16, 32, 64, 128], [32, 64, 128, 256, 512], [2 ** 0, 2 ** (1 / 3), 2 ** (2 / 3)],
                          [1 / 3, 1 / 2, 1, 2, 3])
    pri = O.prior_boxes(table, 64, 64)
    A = pri.shape[0]
    cls = rng.uniform(0, 0.45, (B, A, C)).astype(np.float32)
    hot = rng.random((B, A, C)) < frac
    cls[hot] = (0.5 + 0.5 * (rng.permutation(int(hot.sum())) + 0.5) / max(int(hot.sum()), 1)).astype(np.float32)
    cls[0, 5, 1] = cls[0, 300, 1] = cls[1, 7, 2] = np.float32(0.99)       # equal scores: lower index first on both sides
    boxes = O.restore_boxes((rng.normal(size=(B, A, 4)) * 0.2).astype(np.float32), pri[None])
    for max_out in (100, 7):
        _, kept = O.detection_proposal(cls, boxes, 0.5, 0.4, 0.6, max_out)
        np.testing.assert_array_equal(kept, _detection_proposal_loops(cls, boxes, 0.5, 0.4, 0.6, max_out))


def test_order_stable_fixture_helpers():
    """oracle/fixtures.py: threshold in the widest score gap, logit scaling of the class output kernels, and the
    perturbation check that backs the full-size "indices bit-exact" tests."""
    from oracle import fixtures as FX
    c = np.array([[[0.40], [0.470], [0.471], [0.53], [0.54], [0.9]]], np.float32)
    thr, gap = FX.gap_threshold(c)
    assert 0.471 < thr < 0.53 and abs(gap - (0.53 - 0.471)) < 1e-6
    w = {"classification_sub_net/block0/output/kernel": np.ones((3, 3, 4, 5), np.float32),
         "classification_sub_net/block0/output/bias": np.ones(5, np.float32), "other/kernel": np.ones(3, np.float32)}
    w2 = FX.scale_cls_logits(w, 3.5)
    assert float(w2["classification_sub_net/block0/output/kernel"][0, 0, 0, 0]) == 3.5
    assert w2["classification_sub_net/block0/output/bias"] is w["classification_sub_net/block0/output/bias"]
    assert w2["other/kernel"] is w["other/kernel"]
    from masklab_hip import ModelConfiguration
    cfg = ModelConfiguration()
    pri = FX.boxes_from(cfg, np.zeros((1, 15 * (8 * 8 + 4 * 4 + 2 * 2 + 1 + 1), 4), np.float32), 64, 64)
    cls = np.zeros((1, pri.shape[1], 5), np.float32)
    cls[0, [3, 400, 900], 2] = [0.9, 0.8, 0.7]                  # far apart in score: stable
    kept, same = FX.order_stability(cfg, cls, pri, 0.5, trials=4)
    assert same == 4 and len(kept) >= 1
    cls[0, 401, 2] = np.float32(0.8) + np.float32(1e-6)         # a near-tie between overlapping neighbours: unstable
    _, same = FX.order_stability(cfg, cls, pri, 0.5, trials=16)
    assert same < 16


def test_mobilenet_v1_architecture_vs_transformers_port():
    """SURVEY 8a row a4 / VERDICT "nothing in the tree pins it": the oracle's MobileNet v1 body (restated from
    keras-applications, which the reference calls at engine/backbone/base.py:253-258) against an INDEPENDENT
    implementation -- Hugging Face `transformers.MobileNetV1Model`, a port of the same TF-slim network (TF 'SAME'
    padding, BN eps 1e-3, ReLU6) -- with the same random weights, at an even input size (where keras'
    ZeroPadding2D(((0,1),(0,1))) + 'valid' equals TF 'SAME').  Checks block order, strides, widths, which tensors the
    C1..C5 taps are (conv_pw_1/3/5/11/13), padding side and BN placement."""
    transformers = pytest.importorskip("transformers")
    from transformers import MobileNetV1Config, MobileNetV1Model
    torch.manual_seed(0)
    hf = MobileNetV1Model(MobileNetV1Config(), add_pooling_layer=False).eval()
    g = torch.Generator().manual_seed(1)
    w = {}

    def take(conv_block, name, depthwise):
        k = conv_block.convolution.weight.detach()
        with torch.no_grad():                         # He-style scale so activations stay O(1) through 27 layers
            fan_in = k.shape[1] * k.shape[2] * k.shape[3]
            k.copy_(torch.randn(k.shape, generator=g) * (2.0 / fan_in) ** 0.5)
            bn = conv_block.normalization
            bn.weight.copy_(torch.rand(bn.weight.shape, generator=g) + 0.5)
            bn.bias.copy_(torch.randn(bn.bias.shape, generator=g) * 0.1)
            bn.running_mean.copy_(torch.randn(bn.running_mean.shape, generator=g) * 0.1)
            bn.running_var.copy_(torch.rand(bn.running_var.shape, generator=g) + 0.5)
        kn = k.numpy()
        if depthwise:                                 # torch [C,1,3,3] -> keras depthwise_kernel [3,3,C,1]
            w[f"{name}/depthwise_kernel"] = np.ascontiguousarray(kn.transpose(2, 3, 0, 1))
        else:                                         # torch [O,I,kh,kw] -> keras kernel [kh,kw,I,O]
            w[f"{name}/kernel"] = np.ascontiguousarray(kn.transpose(2, 3, 1, 0))
        for src, dst in (("weight", "gamma"), ("bias", "beta"), ("running_mean", "moving_mean"),
                         ("running_var", "moving_variance")):
            w[f"{name}_bn/{dst}"] = getattr(conv_block.normalization, src).detach().numpy().copy()

    take(hf.conv_stem, "conv1", False)
    assert len(hf.layer) == 26
    for i in range(13):
        take(hf.layer[2 * i], f"conv_dw_{i + 1}", True)
        take(hf.layer[2 * i + 1], f"conv_pw_{i + 1}", False)
    x = np.random.default_rng(2).normal(size=(2, 96, 128, 3)).astype(np.float32)
    taps = O.mobilenet_v1(x.astype(np.float64), w)
    with torch.no_grad():
        out = hf(torch.from_numpy(x).permute(0, 3, 1, 2), output_hidden_states=True)
    hidden = out.hidden_states                       # after each of the 26 layers
    assert len(hidden) == 26
    for name, block in (("C1", 1), ("C2", 3), ("C3", 5), ("C4", 11), ("C5", 13)):
        ref = hidden[2 * block - 1].permute(0, 2, 3, 1).numpy()
        got = taps[name]
        assert got.shape == ref.shape, (name, got.shape, ref.shape)
        assert np.abs(ref).max() > 0.1                # the comparison is not between two all-zero maps
        np.testing.assert_allclose(got, ref, atol=2e-4, err_msg=name)
