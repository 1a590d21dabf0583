"""CPU ORACLE (test infrastructure only) -- the reference's detection quality metric in NumPy.

SURVEY section 8(d): BASELINE.json's "box AP vs Keras ref" has no definition in the reference (F7); the
closest thing it ships is DetectionIOUMetric (engine/metrics.py:109-165): precision / recall / F-measure
of proposed boxes against ground-truth boxes at IoU > 0.5.  bench.py and the GPU tests use it with the
ORACLE's detections as "ground truth", so build-vs-oracle agreement reads 1.0 (up to the metric's own
epsilon).  PARITY UNPINNED like the rest of oracle/ (no TensorFlow here, no reference fixtures)."""
import numpy as np

F32 = np.float32
K_EPSILON = F32(1e-7)            # tf.keras.backend.epsilon()


def calculate_iou(aa_boxes, bb_boxes):
    """CalculateIOU.call, engine/layers/detection.py:391-422.  Boxes (cx, cy, w, h); returns [len(aa), len(bb)]."""
    aa = np.asarray(aa_boxes, F32)[:, :4]
    bb = np.asarray(bb_boxes, F32)[:, :4]
    aa_area = bb[:, 2] * bb[:, 3]                                  # :398 (named the other way round, symmetric)
    bb_area = aa[:, 2] * aa[:, 3]                                  # :399
    areas = aa_area[None, :] + bb_area[:, None]                    # :400

    def corners(b):                                                # NormalizeBoxes without shape (detection.py:362-374)
        cx, cy, w, h = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
        return cy - h / F32(2), cx - w / F32(2), cy + h / F32(2), cx + w / F32(2)

    ay1, ax1, ay2, ax2 = [v[:, None] for v in corners(aa)]
    by1, bx1, by2, bx2 = [v[None, :] for v in corners(bb)]
    in_w = np.maximum(F32(0), np.minimum(bx2, ax2) - np.maximum(bx1, ax1))       # :409-414
    in_h = np.maximum(F32(0), np.minimum(by2, ay2) - np.maximum(by1, ay1))
    inter = in_w * in_h
    return inter / ((areas - inter) + F32(1e-5))                   # :418-421


def detection_iou_metric(proposed_boxes, gt_boxes):
    """DetectionIOUMetric.call, engine/metrics.py:118-160.  Both [B, n, 6] with -1 padded rows.
    Returns per-image (precision, recall, fmeasure)."""
    p = np.asarray(proposed_boxes, F32)
    g = np.asarray(gt_boxes, F32)
    B = p.shape[0]
    prec, rec, fm = [], [], []
    for b in range(B):                                             # :127-147: the batch-diagonal of the big IoU matrix
        ign_p = p[b, :, 0] != -1                                   # :136-137
        ign_g = g[b, :, 0] != -1                                   # :138-139
        mask = np.logical_or(ign_p[:, None], ign_g[None, :]).astype(F32)          # :140-141 (logical_or, as written)
        iou = calculate_iou(p[b], g[b]) * mask                     # :145-147
        num_pos = F32((iou.max(axis=1) > 0.5).sum()) if iou.shape[1] else F32(0)   # :151-152
        num_true = F32((iou.max(axis=0) > 0.5).sum()) if iou.shape[0] else F32(0)  # :153-154
        num_pred = F32(ign_p.sum())                                # :156-157
        num_gt = F32(ign_g.sum())                                  # :158-159
        pr = num_pos / (num_pred + K_EPSILON)                      # :161
        rc = num_true / (num_gt + K_EPSILON)                       # :162
        prec.append(pr)
        rec.append(rc)
        fm.append(F32(2) * (pr * rc) / (pr + rc + K_EPSILON))      # :163
    return np.asarray(prec, F32), np.asarray(rec, F32), np.asarray(fm, F32)
