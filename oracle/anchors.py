"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's RetinaNet
anchor labelling (row f4), used by tests/ to check the HIP implementation.

Follows, line by line:
  detectron/lib/modeling/generate_anchors.py:53-130   (cell anchors)
  detectron/lib/roi_data/data_utils.py:52-103         (field of anchors)
  detectron/lib/roi_data/retinanet.py:75-94           (generate_all_anchors)
  detectron/lib/utils/cython_bbox.pyx:31-74           (bbox_overlaps, float32)
  detectron/lib/roi_data/retinanet.py:198-306         (_get_retinanet_blobs)
  detectron/lib/roi_data/retinanet.py:97-196          (add_retinanet_blobs: stacking)
  detectron/lib/utils/boxes.py:193-224                (bbox_transform_inv)

Pinning: `bbox_overlaps` -- the one numerically delicate piece, because labels
depend on EXACT float32 equality of IoU values -- is checked bit-for-bit against
the reference's own cython source compiled in the build container
(oracle/Makefile target `ref`, tests/test_oracle_golden.py); the labelling logic
itself has no stored answers anywhere in the reference and is restated from the
code ("parity unpinned" for that part).
"""
import numpy as np

F32 = np.float32


class Cfg(object):
    """The configuration values the labelling depends on (core/config.py:92,509-555,
    708-729 with the RetinaNet yaml overrides: levels 3-7, coarsest stride 128)."""
    k_min, k_max = 3, 7
    scales_per_octave = 3
    aspect_ratios = (0.5, 1.0, 2.0)
    anchor_scale = 4
    positive_overlap = 0.5
    negative_overlap = 0.4
    coarsest_stride = 128
    train_max_size = 1000
    num_classes = 81


# ---- generate_anchors.py:53-130 ------------------------------------------------

def _whctrs(anchor):
    w = anchor[2] - anchor[0] + 1
    h = anchor[3] - anchor[1] + 1
    return w, h, anchor[0] + 0.5 * (w - 1), anchor[1] + 0.5 * (h - 1)


def _mkanchors(ws, hs, x_ctr, y_ctr):
    ws = ws[:, np.newaxis]
    hs = hs[:, np.newaxis]
    return np.hstack((x_ctr - 0.5 * (ws - 1), y_ctr - 0.5 * (hs - 1),
                      x_ctr + 0.5 * (ws - 1), y_ctr + 0.5 * (hs - 1)))


def generate_anchors(stride, sizes, aspect_ratios):
    scales = np.array(sizes, dtype=np.float64) / stride
    ratios = np.array(aspect_ratios, dtype=np.float64)
    anchor = np.array([1, 1, stride, stride], dtype=np.float64) - 1
    w, h, x_ctr, y_ctr = _whctrs(anchor)
    size_ratios = (w * h) / ratios
    ws = np.round(np.sqrt(size_ratios))
    hs = np.round(ws * ratios)
    anchors = _mkanchors(ws, hs, x_ctr, y_ctr)
    out = []
    for i in range(anchors.shape[0]):
        w, h, x_ctr, y_ctr = _whctrs(anchors[i, :])
        out.append(_mkanchors(w * scales, h * scales, x_ctr, y_ctr))
    return np.vstack(out)


def cell_anchors(cfg=Cfg):
    """float32 [levels][A][4]: the A = scales_per_octave * len(aspect_ratios) anchors of
    one cell per level, in generate_all_anchors' order (octave outer, aspect inner)."""
    out = []
    for lvl in range(cfg.k_min, cfg.k_max + 1):
        stride = 2.0 ** lvl
        lv = []
        for octave in range(cfg.scales_per_octave):
            octave_scale = 2 ** (octave / float(cfg.scales_per_octave))
            for ar in cfg.aspect_ratios:
                lv.append(generate_anchors(stride, (stride * octave_scale * cfg.anchor_scale,),
                                           (ar,))[0])
        out.append(np.array(lv))
    return np.array(out).astype(F32)


def field_size(stride, cfg=Cfg):
    fpn_max = cfg.coarsest_stride * np.ceil(cfg.train_max_size / float(cfg.coarsest_stride))
    return int(np.ceil(fpn_max / float(stride)))


def all_anchors(cfg=Cfg):
    """The concatenated fields of anchors (float32 [total][4]) and, per field,
    (level index, anchor index, field size): field-major, then y, then x."""
    cells = cell_anchors(cfg)
    fields, meta = [], []
    for li, lvl in enumerate(range(cfg.k_min, cfg.k_max + 1)):
        stride = 2.0 ** lvl
        fs = field_size(stride, cfg)
        shifts = np.arange(0, fs) * stride
        sx, sy = np.meshgrid(shifts, shifts)
        sh = np.vstack((sx.ravel(), sy.ravel(), sx.ravel(), sy.ravel())).transpose()
        for a in range(cells.shape[1]):
            # data_utils.py:89-93: float64 cell anchor + shifts, then astype(float32)
            cell64 = generate_cell64(lvl, a, cfg)
            fields.append((cell64.reshape((1, 4)) + sh).astype(F32))
            meta.append((li, a, fs))
    return np.concatenate(fields), meta


def generate_cell64(lvl, a, cfg=Cfg):
    stride = 2.0 ** lvl
    octave, idx = divmod(a, len(cfg.aspect_ratios))
    octave_scale = 2 ** (octave / float(cfg.scales_per_octave))
    return generate_anchors(stride, (stride * octave_scale * cfg.anchor_scale,),
                            (cfg.aspect_ratios[idx],))[0]


# ---- cython_bbox.pyx:31-74 -----------------------------------------------------

def bbox_overlaps(boxes, query):
    """IoU with the +1 pixel convention, in the arithmetic the cython source compiles
    to: differences of float32 coordinates are float32, but Cython writes the literal
    `1` next to a C float as `1.0` -- a C double -- so `+ 1`, the two area products and
    the union sum run in double and are rounded to float32 only when stored into the
    float32 locals (box_area, iw, ih, ua); iw * ih and the final division are float32."""
    boxes = np.ascontiguousarray(boxes, dtype=F32)
    query = np.ascontiguousarray(query, dtype=F32)
    D = np.float64
    box_area = (((query[:, 2] - query[:, 0]).astype(D) + 1.0) *
                ((query[:, 3] - query[:, 1]).astype(D) + 1.0)).astype(F32)                 # [K]
    iw = ((np.minimum(boxes[:, None, 2], query[None, :, 2]) -
           np.maximum(boxes[:, None, 0], query[None, :, 0])).astype(D) + 1.0).astype(F32)
    ih = ((np.minimum(boxes[:, None, 3], query[None, :, 3]) -
           np.maximum(boxes[:, None, 1], query[None, :, 1])).astype(D) + 1.0).astype(F32)
    area = (((boxes[:, 2] - boxes[:, 0]).astype(D) + 1.0) *
            ((boxes[:, 3] - boxes[:, 1]).astype(D) + 1.0))[:, None]                       # double
    inter = iw * ih                                                                     # float32
    ua = ((area + box_area[None, :].astype(D)) - inter.astype(D)).astype(F32)
    with np.errstate(divide="ignore", invalid="ignore"):
        ov = inter / ua
    return np.where((iw > 0) & (ih > 0), ov, F32(0)).astype(F32)


# ---- boxes.py:193-224 ----------------------------------------------------------

def bbox_transform_inv(boxes, gt):
    boxes = boxes.astype(F32)
    gt = gt.astype(F32)
    ew = boxes[:, 2] - boxes[:, 0] + F32(1.0)
    eh = boxes[:, 3] - boxes[:, 1] + F32(1.0)
    ex = boxes[:, 0] + F32(0.5) * ew
    ey = boxes[:, 1] + F32(0.5) * eh
    gw = gt[:, 2] - gt[:, 0] + F32(1.0)
    gh = gt[:, 3] - gt[:, 1] + F32(1.0)
    gx = gt[:, 0] + F32(0.5) * gw
    gy = gt[:, 1] + F32(0.5) * gh
    return np.vstack(((gx - ex) / ew, (gy - ey) / eh, np.log(gw / ew),
                      np.log(gh / eh))).transpose().astype(F32)


# ---- retinanet.py:198-306 and :97-196 --------------------------------------------

def retinanet_blobs(gt_boxes_list, gt_classes_list, im_height, im_width, cfg=Cfg):
    """gt_boxes_list[i]: float32 [G_i][4] (already scaled), gt_classes_list[i]: int [G_i]
    (1..80); im_height/im_width: the padded blob size.  Returns a dict with, per level
    `lvl`: labels int32 [N][A][h][w], targets float32 [M][4], locs float32 [M][4]
    ([image, 4*anchor, y, x]) in the reference's stacking order, plus fg_num, bg_num."""
    anchors, meta = all_anchors(cfg)
    A = cfg.scales_per_octave * len(cfg.aspect_ratios)
    nlev = cfg.k_max - cfg.k_min + 1
    labels_out = [[] for _ in range(nlev)]
    targets_out = [[] for _ in range(nlev)]
    locs_out = [[] for _ in range(nlev)]
    fg_total, bg_total = 0.0, 0.0
    for im_i, (gt_boxes, gt_classes) in enumerate(zip(gt_boxes_list, gt_classes_list)):
        gt_boxes = np.asarray(gt_boxes, dtype=F32)
        gt_classes = np.asarray(gt_classes)
        total = anchors.shape[0]
        labels = np.empty((total,), dtype=F32)
        labels.fill(-1)
        ov = bbox_overlaps(anchors, gt_boxes)
        a2g_argmax = ov.argmax(axis=1)
        a2g_max = ov[np.arange(total), a2g_argmax]
        g2a_argmax = ov.argmax(axis=0)
        g2a_max = ov[g2a_argmax, np.arange(ov.shape[1])]
        with_max = np.where(ov == g2a_max)[0]
        labels[with_max] = gt_classes[a2g_argmax[with_max]]
        inds = a2g_max >= F32(cfg.positive_overlap)
        labels[inds] = gt_classes[a2g_argmax[inds]]
        fg_inds = np.where(labels >= 1)[0]
        bg_inds = np.where(a2g_max < F32(cfg.negative_overlap))[0]
        labels[bg_inds] = 0
        num_fg, num_bg = len(fg_inds), len(bg_inds)
        bbox_targets = np.zeros((total, 4), dtype=F32)
        bbox_targets[fg_inds, :] = bbox_transform_inv(anchors[fg_inds, :],
                                                      gt_boxes[a2g_argmax[fg_inds], :])
        start = 0
        per_level_labels = [[] for _ in range(nlev)]
        for (li, a, fs) in meta:
            end = start + fs * fs
            _labels = labels[start:end].reshape((1, 1, fs, fs))
            _targets = bbox_targets[start:end, :].reshape((1, fs, fs, 4)).transpose(0, 3, 1, 2)
            start = end
            stride = 2.0 ** (cfg.k_min + li)
            w = int(im_width / stride)
            h = int(im_height / stride)
            # NB (reference behaviour): the fg list is taken from the WHOLE field, not the
            # cropped one (retinanet.py:278-293 use _labels before the [0:h, 0:w] crop)
            ys, xs = np.where(_labels[0, 0] > 0)
            for y, x in zip(ys, xs):
                targets_out[li].append(_targets[0, :, y, x])
                locs_out[li].append(np.array([im_i, 4 * a, y, x], dtype=F32))
            per_level_labels[li].append(_labels[:, :, 0:h, 0:w].astype(np.int32))
        for li in range(nlev):
            labels_out[li].append(np.concatenate(per_level_labels[li], axis=1))
        fg_total += float(num_fg)
        bg_total += (num_bg + 1.0) * (cfg.num_classes - 1) + float(num_fg) * (cfg.num_classes - 2)
    out = {"fg_num": F32(fg_total), "bg_num": F32(bg_total)}
    for li in range(nlev):
        lvl = cfg.k_min + li
        out["labels_fpn%d" % lvl] = np.concatenate(labels_out[li], axis=0)
        out["targets_fpn%d" % lvl] = (np.array(targets_out[li], dtype=F32).reshape(-1, 4))
        out["locs_fpn%d" % lvl] = (np.array(locs_out[li], dtype=F32).reshape(-1, 4))
    return out
