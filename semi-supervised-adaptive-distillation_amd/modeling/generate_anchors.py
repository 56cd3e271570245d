"""Cell anchors and anchor-field geometry of RetinaNet (host side, a few hundred flops).

Mirrors detectron/lib/modeling/generate_anchors.py:53-130 (anchors are centred
on stride/2 with rounded widths/heights per aspect ratio, then scaled),
detectron/lib/roi_data/retinanet.py:75-94 (one anchor per (octave, aspect ratio),
size = stride * 2^(octave/3) * ANCHOR_SCALE) and data_utils.py:71-75 (the field
of anchors spans COARSEST_STRIDE * ceil(MAX_SIZE / COARSEST_STRIDE) pixels)."""
import numpy as np


class AnchorConfig(object):
    k_min, k_max = 3, 7                  # FPN.RPN_MIN_LEVEL / RPN_MAX_LEVEL (RetinaNet yaml)
    scales_per_octave = 3                # RETINANET.SCALES_PER_OCTAVE
    aspect_ratios = (0.5, 1.0, 2.0)      # RETINANET.ASPECT_RATIOS
    anchor_scale = 4                     # RETINANET.ANCHOR_SCALE
    positive_overlap = 0.5               # RETINANET.POSITIVE_OVERLAP
    negative_overlap = 0.4               # RETINANET.NEGATIVE_OVERLAP
    coarsest_stride = 128                # FPN.COARSEST_STRIDE
    train_max_size = 1000                # TRAIN.MAX_SIZE
    num_classes = 81                     # MODEL.NUM_CLASSES


def _whctrs(a):
    w = a[2] - a[0] + 1
    h = a[3] - a[1] + 1
    return w, h, a[0] + 0.5 * (w - 1), a[1] + 0.5 * (h - 1)


def _mk(w, h, xc, yc):
    return np.array([xc - 0.5 * (w - 1), yc - 0.5 * (h - 1), xc + 0.5 * (w - 1), yc + 0.5 * (h - 1)])


def cell_anchor(stride, size, aspect_ratio):
    """The single anchor of generate_anchors(stride, (size,), (aspect_ratio,)), float64."""
    w, h, xc, yc = _whctrs(np.array([0.0, 0.0, stride - 1.0, stride - 1.0]))
    ws = np.round(np.sqrt(w * h / aspect_ratio))
    hs = np.round(ws * aspect_ratio)
    w, h, xc, yc = _whctrs(_mk(ws, hs, xc, yc))
    scale = float(size) / stride
    return _mk(w * scale, h * scale, xc, yc)


def cell_anchors(cfg=AnchorConfig):
    """float64 [levels][A][4], octave-major then aspect ratio (generate_all_anchors order)."""
    out = []
    for lvl in range(cfg.k_min, cfg.k_max + 1):
        stride = 2.0 ** lvl
        out.append([cell_anchor(stride, stride * 2 ** (o / float(cfg.scales_per_octave)) *
                                cfg.anchor_scale, ar)
                    for o in range(cfg.scales_per_octave) for ar in cfg.aspect_ratios])
    return np.array(out, dtype=np.float64)


def field_sizes(cfg=AnchorConfig):
    fpn_max = cfg.coarsest_stride * np.ceil(cfg.train_max_size / float(cfg.coarsest_stride))
    return [int(np.ceil(fpn_max / float(2 ** lvl))) for lvl in range(cfg.k_min, cfg.k_max + 1)]
