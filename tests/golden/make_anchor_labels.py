#!/usr/bin/env python3
"""Fixtures for anchor labelling and box decoding, produced BY THE REFERENCE'S OWN PYTHON
(SURVEY.md 8 row f4): detectron/lib/roi_data/retinanet.py:97-306 (add_retinanet_blobs /
_get_retinanet_blobs), roi_data/data_utils.py:50-103, modeling/generate_anchors.py and, for the
decode arithmetic, utils/boxes.py (bbox_transform, clip_tiled_boxes) -- imported from
/root/reference under python 3 with the sys.modules stubs of make_head_graph.py, the reference's
IoU routine compiled from its own cython_bbox.pyx (oracle/_pyref/cython_bbox.so, `make -C oracle
pyref`; build container only, never shipped), cfg fields set by assignment.  utils/cython_nms.pyx cannot be built here (numpy 2's
Cython declarations have no `int_t`), so greedy NMS stays unpinned; nothing in this fixture goes
through it.

Runs ONLY in the build container.  Output: tests/golden/anchor_labels_ref.npz (data only: the
synthetic ground truth that went in, every blob that came out).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_anchor_labels.py
"""
import importlib.util
import os
import sys
import types
from collections import defaultdict

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/detectron/lib"
OUT = os.path.join(HERE, "anchor_labels_ref.npz")


def install():
    sys.path.insert(0, HERE)
    import make_head_graph as mh
    mh.install_stubs()
    sys.path.insert(0, REF)
    # the reference's compiled IoU (cython_bbox.pyx -> oracle/_pyref/cython_bbox.so)
    so = os.path.join(ROOT, "oracle", "_pyref", "cython_bbox.so")
    spec = importlib.util.spec_from_file_location("utils.cython_bbox", so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    sys.modules["utils.cython_bbox"] = mod
    sys.modules["utils.cython_nms"] = types.ModuleType("utils.cython_nms")     # unbuildable here; unused below
    # the builtin-type aliases numpy >= 1.24 removed (the reference's numpy code spells np.float, np.int)
    for alias, typ in (("int", int), ("float", float), ("bool", bool)):
        if alias not in np.__dict__:
            setattr(np, alias, typ)


def main():
    install()
    from core.config import cfg
    import utils.boxes as box_utils
    import roi_data.retinanet as rn
    # configs/focal_distillation/retinanet_R-50-FPN_distillation.yaml
    cfg.FPN.FPN_ON = True
    cfg.FPN.MULTILEVEL_RPN = True
    cfg.FPN.RPN_MAX_LEVEL, cfg.FPN.RPN_MIN_LEVEL = 7, 3
    cfg.FPN.COARSEST_STRIDE = 128
    cfg.RETINANET.RETINANET_ON = True
    cfg.RETINANET.SCALES_PER_OCTAVE = 3
    cfg.RETINANET.ASPECT_RATIOS = (0.5, 1.0, 2.0)
    cfg.RETINANET.ANCHOR_SCALE = 4
    cfg.RETINANET.POSITIVE_OVERLAP, cfg.RETINANET.NEGATIVE_OVERLAP = 0.5, 0.4
    cfg.RETINANET.CLASS_SPECIFIC_BBOX = False
    cfg.MODEL.NUM_CLASSES = 81
    cfg.TRAIN.MAX_SIZE = 1000

    rng = np.random.default_rng(20260929)
    image_height, image_width = 640, 896          # the padded blob (a 600 px image)
    roidb, scales = [], []
    for n_gt, (h, w) in ((7, (427, 640)), (12, (480, 640))):
        scale = 600.0 / min(h, w)
        if np.round(scale * max(h, w)) > 1000:
            scale = 1000.0 / max(h, w)
        boxes = np.zeros((n_gt, 4), np.float32)
        for g in range(n_gt):
            # log-uniform sizes: every pyramid level gets foreground anchors
            bw = float(np.exp(rng.uniform(np.log(14.0), np.log(0.7 * w))))
            bh = float(np.clip(bw * np.exp(rng.uniform(-0.7, 0.7)), 10.0, 0.9 * h))
            x1, y1 = rng.uniform(0, w - bw - 1), rng.uniform(0, h - bh - 1)
            boxes[g] = [x1, y1, x1 + bw, y1 + bh]
        roidb.append(dict(height=h, width=w, boxes=boxes,
                          gt_classes=rng.integers(1, 81, size=n_gt).astype(np.int32),
                          is_crowd=np.zeros(n_gt, np.int32)))
        scales.append(scale)
    blobs = defaultdict(list)
    ok = rn.add_retinanet_blobs(blobs, scales, roidb, image_width, image_height)
    assert ok
    out = {"image_hw": np.array([image_height, image_width], np.int32), "scales": np.array(scales, np.float64)}
    for i, e in enumerate(roidb):
        out["gt_boxes_%d" % i] = e["boxes"]
        out["gt_classes_%d" % i] = e["gt_classes"]
        out["orig_hw_%d" % i] = np.array([e["height"], e["width"]], np.int32)
    for k, v in blobs.items():
        out["blob_" + k] = np.asarray(v)
    # the anchors the reference enumerated (one field per (level, octave, aspect))
    foas = rn.generate_all_anchors()
    out["cell_anchors"] = np.stack([f.field_of_anchors[0] for f in foas]).astype(np.float32)
    out["field_sizes"] = np.array([f.field_size for f in foas], np.int32)
    # decode arithmetic (utils/boxes.py:193-260): deltas -> boxes, clip to the image
    anchors = np.concatenate([f.field_of_anchors[:64] for f in foas[::9]]).astype(np.float32)
    deltas = (rng.standard_normal((anchors.shape[0], 4)) * np.array([0.3, 0.3, 0.6, 0.6])).astype(np.float32)
    deltas[::17, 2:] = 6.0                        # beyond cfg.BBOX_XFORM_CLIP: the clamp is exercised
    pred = box_utils.bbox_transform(anchors, deltas)
    out["decode_anchors"], out["decode_deltas"] = anchors, deltas
    out["decode_boxes"] = pred.astype(np.float32)
    out["decode_boxes_dtype_was"] = np.array(str(pred.dtype))
    out["decode_clipped"] = box_utils.clip_tiled_boxes(pred.copy(), (600, 899, 3)).astype(np.float32)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: getattr(v, "shape", None) for k, v in out.items() if k.startswith("blob_")})


if __name__ == "__main__":
    main()
