"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's RetinaNet
inference post-processing (row f4, second half), used by tests/ to check the HIP
implementation.

Follows detectron/lib/core/test_retinanet.py:108-206 (per-level score threshold,
top-k, anchor decode, clipping, per-class NMS, final top-N), utils/boxes.py:132-190
(clip_tiled_boxes, bbox_transform) and utils/cython_nms.pyx:37-92 (greedy NMS).
Pinned where the reference can be run: box decoding / clipping by fixtures of the imported utils/boxes.py
(tests/golden/anchor_labels_ref.npz), greedy NMS by the survivors that cython_nms.pyx:37-92 itself leaves when
its text is executed as Python after stripping the C type declarations (tests/golden/make_nms_golden.py ->
nms_ref.npz; the .pyx does not build here: it uses `np.int_t` / `np.int`, which Cython 3 / numpy 2 no longer
provide).  PARITY UNPINNED for the rest: the per-level threshold / top-k / final top-N composition of
test_retinanet.py is restated from the code (it needs a Caffe2 workspace; the reference holds no stored
detections).  argpartition / argsort tie orders are unspecified in the reference; tests use distinct scores.
"""
import numpy as np

F32 = np.float32
BBOX_XFORM_CLIP = np.log(1000. / 16.)      # core/config.py:923


def bbox_transform(boxes, deltas):
    """boxes.py:150-190 with weights (1, 1, 1, 1), float32."""
    if boxes.shape[0] == 0:
        return np.zeros((0, deltas.shape[1]), dtype=deltas.dtype)
    boxes = boxes.astype(deltas.dtype, copy=False)
    widths = boxes[:, 2] - boxes[:, 0] + F32(1.0)
    heights = boxes[:, 3] - boxes[:, 1] + F32(1.0)
    ctr_x = boxes[:, 0] + F32(0.5) * widths
    ctr_y = boxes[:, 1] + F32(0.5) * heights
    dx, dy = deltas[:, 0::4], deltas[:, 1::4]
    dw = np.minimum(deltas[:, 2::4], F32(BBOX_XFORM_CLIP))
    dh = np.minimum(deltas[:, 3::4], F32(BBOX_XFORM_CLIP))
    pcx = dx * widths[:, None] + ctr_x[:, None]
    pcy = dy * heights[:, None] + ctr_y[:, None]
    pw = np.exp(dw) * widths[:, None]
    ph = np.exp(dh) * heights[:, None]
    out = np.zeros(deltas.shape, dtype=deltas.dtype)
    out[:, 0::4] = pcx - F32(0.5) * pw
    out[:, 1::4] = pcy - F32(0.5) * ph
    out[:, 2::4] = pcx + F32(0.5) * pw - F32(1)
    out[:, 3::4] = pcy + F32(0.5) * ph - F32(1)
    return out


def clip_tiled_boxes(boxes, im_shape):
    boxes[:, 0::4] = np.maximum(np.minimum(boxes[:, 0::4], im_shape[1] - 1), 0)
    boxes[:, 1::4] = np.maximum(np.minimum(boxes[:, 1::4], im_shape[0] - 1), 0)
    boxes[:, 2::4] = np.maximum(np.minimum(boxes[:, 2::4], im_shape[1] - 1), 0)
    boxes[:, 3::4] = np.maximum(np.minimum(boxes[:, 3::4], im_shape[0] - 1), 0)
    return boxes


def nms(dets, thresh):
    """cython_nms.pyx:37-92: float32 throughout, suppression at ovr >= thresh, result =
    indices of the survivors in their ORIGINAL order."""
    dets = np.ascontiguousarray(dets, dtype=F32)
    x1, y1, x2, y2, scores = (dets[:, i] for i in range(5))
    areas = (x2 - x1 + F32(1)) * (y2 - y1 + F32(1))
    order = scores.argsort()[::-1]
    n = dets.shape[0]
    suppressed = np.zeros(n, dtype=np.int64)
    thresh = F32(thresh)
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        rest = order[_i + 1:]
        rest = rest[suppressed[rest] == 0]
        xx1 = np.maximum(x1[i], x1[rest])
        yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest])
        yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(F32(0.0), xx2 - xx1 + F32(1))
        h = np.maximum(F32(0.0), yy2 - yy1 + F32(1))
        inter = w * h
        ovr = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[ovr >= thresh]] = 1
    return np.where(suppressed == 0)[0]


def im_detect_bbox(cls_probs, box_preds, cell_anchors, im_shape, scale, k_min=3,
                   num_classes=81, inference_th=0.05, pre_nms_topn=1000, nms_thresh=0.5,
                   dets_per_im=100):
    """cls_probs[l]: float32 [1][A*C][H][W] (sigmoid scores), box_preds[l]: [1][A*4][H][W],
    cell_anchors[l]: float64 [A][4].  Returns float32 [n <= dets_per_im][6] =
    x1, y1, x2, y2, score, class, sorted by score (test_retinanet.py:108-206)."""
    k_max = k_min + len(cls_probs) - 1
    boxes_all = {}
    for li, lvl in enumerate(range(k_min, k_max + 1)):
        stride = 2. ** lvl
        ca = cell_anchors[li]
        A = ca.shape[0]
        cls_prob = cls_probs[li]
        box_pred = box_preds[li]
        cls_prob = cls_prob.reshape((cls_prob.shape[0], A, int(cls_prob.shape[1] / A),
                                     cls_prob.shape[2], cls_prob.shape[3]))
        box_pred = box_pred.reshape((box_pred.shape[0], A, 4, box_pred.shape[2], box_pred.shape[3]))
        ravel = cls_prob.ravel()
        th = inference_th if lvl < k_max else 0.0
        cand = np.where(ravel > th)[0]
        if len(cand) == 0:
            continue
        k = min(pre_nms_topn, len(cand))
        inds = np.argpartition(ravel[cand], -k)[-k:]
        inds = cand[inds]
        i5 = np.array(np.unravel_index(inds, cls_prob.shape)).transpose()
        classes = i5[:, 2]
        a_ids, y, x = i5[:, 1], i5[:, 3], i5[:, 4]
        scores = cls_prob[0, a_ids, classes, y, x]
        boxes = np.column_stack((x, y, x, y)).astype(dtype=F32)
        boxes *= stride
        boxes += ca[a_ids, :]
        deltas = box_pred[0, a_ids, :, y, x]
        pred = bbox_transform(boxes, deltas)
        pred /= F32(scale)
        pred = clip_tiled_boxes(pred, im_shape)
        bs = np.zeros((pred.shape[0], 5))
        bs[:, 0:4] = pred
        bs[:, 4] = scores
        for cls in range(1, num_classes):
            sel = np.where(classes == cls - 1)[0]
            if len(sel) > 0:
                boxes_all.setdefault(cls, []).extend(bs[sel, :])
    dets = []
    for cls, boxes in boxes_all.items():
        cd = np.vstack(boxes).astype(dtype=F32)
        keep = nms(cd, nms_thresh)
        cd = cd[keep, :]
        out = np.zeros((len(keep), 6))
        out[:, 0:5] = cd
        out[:, 5].fill(cls)
        dets.append(out)
    if not dets:
        return np.zeros((0, 6), F32)
    dets = np.vstack(dets)
    order = np.argsort(-dets[:, 4])
    return dets[order[0:dets_per_im], :].astype(F32)
