"""GPU parity tests for the detection post-process kernels (index work: bit-exact) and the
RoI crop; -m gpu."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import masklab as O
from oracle import tfops as T

RNG = np.random.default_rng(23)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    torch.cuda.synchronize()
    return t.cpu().numpy()


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _anchors(H, W):
    table = O.prior_table([8, 16, 32, 64, 128], [32, 64, 128, 256, 512],
                          [2 ** 0, 2 ** (1 / 3), 2 ** (2 / 3)], [1 / 3, 1 / 2, 1, 2, 3])
    return O.prior_boxes(table, H, W)


def _synthetic_head(B, H, W, C=5, frac=0.01, seed=0):
    """cls scores with `frac` of entries above 0.5 (no ties), loc deltas ~ N(0, 0.2)."""
    rng = np.random.default_rng(seed)
    pri = _anchors(H, W)
    A = pri.shape[0]
    cls = rng.uniform(0.0, 0.45, (B, A, C)).astype(np.float32)
    hot = rng.random((B, A, C)) < frac
    n_hot = int(hot.sum())
    # distinct scores by construction (TF leaves the order of equal scores unspecified)
    cls[hot] = (0.5 + 0.5 * (rng.permutation(n_hot) + 0.5) / max(n_hot, 1)).astype(np.float32)
    assert len(np.unique(cls[hot])) == n_hot
    loc = (rng.normal(size=(B, A, 4)) * 0.2).astype(np.float32)
    return pri, cls, loc


def test_restore_boxes():
    from masklab_hip import ops
    pri, _, loc = _synthetic_head(2, 128, 128)
    ref = O.restore_boxes(loc, pri[None])
    got = host(ops.restore_boxes(dev(loc), dev(pri)))
    np.testing.assert_allclose(got, ref, rtol=2e-7, atol=1e-5)
    np.testing.assert_array_equal(got[..., :2], ref[..., :2])      # mul+add, no contraction: exact


@pytest.mark.parametrize("B,hw,frac,thr", [(1, (128, 128), 0.01, 0.5), (3, (128, 256), 0.02, 0.5),
                                           (2, (256, 256), 0.002, 0.5), (2, (128, 128), 0.3, 0.5),
                                           (2, (128, 128), 0.0, 0.5), (4, (128, 128), 0.01, 0.05)])
def test_detection_proposal_bit_exact(B, hw, frac, thr):
    from masklab_hip import ops
    pri, cls, loc = _synthetic_head(B, hw[0], hw[1], frac=frac, seed=B)
    if frac == 0.0:
        cls[1:, :, :] = np.minimum(cls[1:], 0.4)
        cls[0, 100, 2] = 0.9            # image 0 has one detection, the others none
    boxes = O.restore_boxes(loc, pri[None])
    ref, kept_ref = O.detection_proposal(cls, boxes, thr, 0.4, 0.6, 100)
    prop, counts, kept = ops.detection_proposal(dev(cls), dev(boxes), thr, 0.4, 0.6, 100, want_kept=True)
    prop, counts, kept = host(prop), host(counts), host(kept)
    n = max(1, int(counts.max()))
    assert ref.shape == (B, n, 6)
    np.testing.assert_array_equal(prop[:, :n], ref)                # rows incl. -1 padding, bit exact
    assert np.all(prop[:, n:] == -1)
    for b in range(B):
        want = kept_ref[kept_ref[:, 0] == b][:, 1:]
        np.testing.assert_array_equal(kept[b, :counts[b]], want)   # (anchor, class) indices
        assert np.all(kept[b, counts[b]:] == -1)


def _check_single_image(cls, boxes, nms_iou, post_iou):
    from masklab_hip import ops
    ref, kept_ref = O.detection_proposal(cls, boxes, 0.5, nms_iou, post_iou, 100)
    prop, counts, kept = ops.detection_proposal(dev(cls), dev(boxes), 0.5, nms_iou, post_iou, 100, want_kept=True)
    prop, counts, kept = host(prop), host(counts), host(kept)
    np.testing.assert_array_equal(kept[0, :counts[0]], kept_ref[:, 1:])
    np.testing.assert_array_equal(prop[:, :ref.shape[1]], ref)
    return int(counts[0])


def test_detection_proposal_many_equal_scores_falls_back_exactly():
    """> 4096 candidates with the SAME score in one (image,class) bucket: one histogram bin overflows the
    LDS band, so the kernel takes its global-memory rounds for that bin; the result still equals the
    oracle's (lower index first among equal scores) and the higher-scored band before it is honoured."""
    pri = _anchors(256, 256)
    A = pri.shape[0]
    cls = np.zeros((1, A, 5), np.float32)
    rng = np.random.default_rng(3)
    same = rng.choice(A, 6000, replace=False)
    cls[0, same, 2] = 0.625
    hi = rng.choice(A, 50, replace=False)
    cls[0, hi, 2] = (0.7 + 0.25 * (rng.permutation(50) + 0.5) / 50).astype(np.float32)
    loc = (rng.normal(size=(1, A, 4)) * 0.2).astype(np.float32)
    boxes = O.restore_boxes(loc, pri[None])
    assert _check_single_image(cls, boxes, 0.4, 0.6) > 20


def test_detection_proposal_multi_band_bucket():
    """one bucket with 20 k distinct-score candidates (five LDS bands) and weak suppression by IoU but
    strong suppression by duplicates: 3 of 4 candidates are exact copies of a higher-scored box, so the
    first band yields < 100 picks and later bands are reached with pre-suppression by earlier picks."""
    pri = _anchors(512, 512)
    A = pri.shape[0]
    rng = np.random.default_rng(8)
    cls = np.zeros((1, A, 5), np.float32)
    idx = rng.choice(A, 20000, replace=False)
    cls[0, idx, 1] = (0.5 + 0.5 * (rng.permutation(20000) + 0.5) / 20000).astype(np.float32)
    boxes = O.restore_boxes(np.zeros((1, A, 4), np.float32), pri[None]).copy()
    # collapse the candidates onto 60 distinct boxes => at most 60 survivors, found across several bands
    # 40 of them only among the 15 000 best scores, the other 20 only below (=> picks from a late band)
    proto = boxes[0, rng.choice(A, 60, replace=False)]
    by_score = idx[np.argsort(-cls[0, idx, 1], kind='stable')]
    boxes[0, by_score[:15000]] = proto[rng.integers(0, 40, 15000)]
    boxes[0, by_score[15000:]] = proto[rng.integers(40, 60, 5000)]
    assert _check_single_image(cls, boxes, 0.95, 0.99) == 60


def test_detection_proposal_tie_break_lower_index():
    from masklab_hip import ops
    pri = _anchors(128, 128)
    A = pri.shape[0]
    cls = np.zeros((1, A, 5), np.float32)
    cls[0, [10, 500, 3000], 1] = 0.75          # three equal scores, far apart (no suppression)
    cls[0, 4000, 3] = 0.75
    loc = np.zeros((1, A, 4), np.float32)
    boxes = O.restore_boxes(loc, pri[None])
    ref, kept_ref = O.detection_proposal(cls, boxes, 0.5, 0.4, 0.6, 100)
    prop, counts, kept = ops.detection_proposal(dev(cls), dev(boxes), 0.5, 0.4, 0.6, 100, want_kept=True)
    np.testing.assert_array_equal(host(kept)[0, :host(counts)[0]], kept_ref[:, 1:])
    np.testing.assert_array_equal(host(prop)[:, :ref.shape[1]], ref)


@pytest.mark.parametrize("B,hw,frac", [(2, (128, 128), 0.01), (1, (256, 256), 0.05), (3, (128, 128), 0.3)])
def test_detection_proposal_keras_default_max_output_1000(B, hw, frac):
    """nms_max_output_size=1000 is the DetectionProposal constructor default (reference detection.py:472):
    C*max_out = 5000 candidates per image do not fit the LDS cross-class stage, so it runs through the
    workspace (gather -> banded bucket NMS -> rows).  Same bit-exact bar; up to 1000 rows come out."""
    from masklab_hip import ops
    pri, cls, loc = _synthetic_head(B, hw[0], hw[1], frac=frac, seed=10 + B)
    boxes = O.restore_boxes(loc, pri[None])
    ref, kept_ref = O.detection_proposal(cls, boxes, 0.5, 0.4, 0.6, 1000)
    prop, counts, kept = ops.detection_proposal(dev(cls), dev(boxes), 0.5, 0.4, 0.6, 1000, want_kept=True)
    prop, counts, kept = host(prop), host(counts), host(kept)
    n = max(1, int(counts.max()))
    assert ref.shape == (B, n, 6) and (frac < 0.05 or n > 100)     # the dense case really exceeds the old cap
    np.testing.assert_array_equal(prop[:, :n], ref)
    assert np.all(prop[:, n:] == -1)
    for b in range(B):
        np.testing.assert_array_equal(kept[b, :counts[b]], kept_ref[kept_ref[:, 0] == b][:, 1:])


def test_detection_gather_payload_is_proposed_plus_count():
    """gather_payload (the all-gather record): per image the 600 floats of `proposed` and the count bit-cast."""
    from masklab_hip import ops, parallel
    pri, cls, loc = _synthetic_head(3, 128, 128, frac=0.01, seed=4)
    boxes = O.restore_boxes(loc, pri[None])
    prop, counts, _, payload = ops.detection_proposal(dev(cls), dev(boxes), 0.5, 0.4, 0.6, 100, want_payload=True)
    p2, c2 = parallel.unpack_payload(payload, 100)
    torch.cuda.synchronize()
    assert payload.shape == (3, 601)
    assert torch.equal(p2, prop) and torch.equal(c2, counts)
    assert p2.data_ptr() == payload.data_ptr()                     # views, not copies


def test_mask_distribute_and_roi_crop():
    from masklab_hip import ops
    from masklab_hip.layers import MaskDistribute, PyramidRoiAlign
    B, H, W = 3, 256, 256
    rng = np.random.default_rng(5)
    n_real = [7, 0, 12]
    cap = 12
    prop = np.full((B, cap, 6), -1.0, np.float32)
    for b, n in enumerate(n_real):
        cx, cy = rng.uniform(20, 236, n), rng.uniform(20, 236, n)
        w, h = rng.uniform(10, 300, n), rng.uniform(10, 300, n)   # some boxes leave the image
        prop[b, :n] = np.stack([cx, cy, w, h, rng.integers(0, 5, n), rng.uniform(0.5, 1, n)], 1)
    fmaps = [rng.normal(size=(B, H // s, W // s, 128)).astype(np.float32) for s in (8, 16, 32)]
    dist_ref = O.mask_distribute(prop, 2, 36)
    rf_ref, rb_ref = O.pyramid_roi_align(fmaps, dist_ref, (H, W), (14, 14))
    dist = MaskDistribute(max_k=2, base_size=36)(dev(prop))
    np.testing.assert_array_equal(host(dist), dist_ref)
    roi_fmaps, roi_boxes = PyramidRoiAlign((14, 14))([[dev(f) for f in fmaps], dist, torch.zeros(B, H, W, 3)])
    np.testing.assert_array_equal(host(roi_boxes), rb_ref)
    for g, r in zip(roi_fmaps, rf_ref):
        g = host(g)
        assert g.shape == r.shape
        np.testing.assert_array_equal(g == -1.0, r == -1.0)        # MoldBatch padding pattern
        np.testing.assert_allclose(g, r, atol=2e-5)
        np.testing.assert_array_equal(g == 0.0, r == 0.0)          # extrapolation cells (in_y<0 etc.)
    # the per-level maxima the kernel reports (the host's one read) = max over images of the level counts
    slots, lcounts, lmax, _ = ops.mask_distribute(dev(prop), 2, 36.0)
    np.testing.assert_array_equal(host(lmax), host(lcounts).max(axis=0))
    np.testing.assert_array_equal(host(lcounts).sum(axis=1), n_real)
    # fused path used by the model: k computed inside from the [B,cap,6] proposals
    rf2, rb2 = PyramidRoiAlign((14, 14)).crop_levels([dev(f) for f in fmaps], dev(prop), (H, W), has_k=False,
                                                      base_size=36)
    np.testing.assert_array_equal(host(rb2), rb_ref)
    for g, r in zip(rf2, rf_ref):
        np.testing.assert_allclose(host(g), r, atol=2e-5)


def test_prior_layer_matches_oracle():
    from masklab_hip import PriorBoxes
    from masklab_hip.layers import PriorLayer
    pb = PriorBoxes([8, 16, 32, 64, 128], [32, 64, 128, 256, 512], [2 ** 0, 2 ** (1 / 3), 2 ** (2 / 3)],
                    [1 / 3, 1 / 2, 1, 2, 3])
    out = PriorLayer(pb)(torch.zeros(2, 128, 384, 3, device="cuda"))
    ref = _anchors(128, 384)
    assert out.shape == (2,) + ref.shape and out.dtype == torch.int32
    np.testing.assert_array_equal(host(out[1].contiguous()), ref)


@pytest.mark.parametrize("B,L,cap,tail,lmax", [(1, 3, 100, (28, 28, 5), [37, 0, 12]), (2, 3, 10, (6,), [10, 1, 3]),
                                               (3, 4, 7, (4, 4, 3), [0, 0, 0, 0]), (2, 2, 5, (6,), [9, 2])])
def test_mold_levels_on_the_device_equals_the_host_sized_concatenation(B, L, cap, tail, lmax):
    """ml_mold_levels_dev_f32 (round 4): MoldBatch + Concatenate(axis=1) of a fixed-capacity stage 2 with the level sizes
    read on the DEVICE -- n_l = min(max(1, lmax), cap), the host rule (reference engine/layers/misc.py:231-286,
    instance.py:222-225) -- so that the launch can be part of the captured forward.  The front of its capacity buffer, taken
    as a view once the host knows n_l, equals the slicing the reference's layers do; both row lengths (E % 4 == 0: 16-byte
    copies; E = 6: the box rows)."""
    from masklab_hip import ops
    rng = np.random.default_rng(3)
    src = rng.normal(size=(B, L * cap) + tail).astype(np.float32)
    n_l = [min(max(1, v), cap) for v in lmax]
    want = np.concatenate([src[:, l * cap:l * cap + n] for l, n in enumerate(n_l)], axis=1)
    buf = ops.mold_levels_dev(dev(src), dev(np.asarray(lmax, np.int32)), cap)
    got = ops.molded_front(buf, n_l)
    assert got.is_contiguous() and tuple(got.shape) == want.shape
    np.testing.assert_array_equal(host(got), want)
    if int(np.prod(tail)) % 4 == 0:            # the host-sized launch (kept in the ABI) gives the same tensor
        np.testing.assert_array_equal(host(ops.mold_levels(dev(src), n_l, cap)), want)
