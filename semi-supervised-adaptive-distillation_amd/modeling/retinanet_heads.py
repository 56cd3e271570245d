"""RetinaNet subnet + adaptive-distillation graph builder.

Emits, through the caffe2_hip Net API, the operator list the reference builds
in detectron/lib/modeling/retinanet_heads.py:63-245 (towers + prediction
convs, level k_min owns the weights and the other levels share them),
:259-307 (SelectSmoothL1Loss / SigmoidFocalLoss) and :313-352 (PowSum +
SigmoidAdaptiveDistillLoss), with the same blob names, argument names and
values -- tests/test_head_graph.py checks it op by op against the list
captured from the reference (tests/golden/head_graph_r50_distill.json).

Configuration is a plain dataclass instead of the reference's global cfg.
"""
from dataclasses import dataclass, field
from typing import Tuple

import numpy as np

from ..caffe2_hip import core


@dataclass
class HeadConfig:
    """Values of configs/focal_distillation/retinanet_R-50-FPN_distillation.yaml
    and the RETINANET / DISTILLATION defaults of core/config.py:988-1016."""
    num_classes: int = 81                 # incl. background
    num_gpus: int = 8
    fpn_dim: int = 256
    k_min: int = 3
    k_max: int = 7
    num_convs: int = 4
    aspect_ratios: Tuple[float, ...] = (1.0, 2.0, 0.5)
    scales_per_octave: int = 3
    prior_prob: float = 0.01
    focal_gamma: float = 2.0
    focal_alpha: float = 0.25
    bbox_reg_beta: float = 0.11
    bbox_reg_weight: float = 1.0
    distill_alpha: float = 0.5
    distill_gamma: float = 2.0
    distill_beta: float = 0.0
    ignored_label: int = -1
    adaptive_normalizer: bool = True
    logits_power: float = 1.8
    temperature: float = 1.0
    use_cudnn_engine: bool = True         # the reference asks for engine=CUDNN
    fuse_relu: bool = False               # extension: fold Relu into the conv

    @property
    def num_anchors(self):
        return len(self.aspect_ratios) * self.scales_per_octave

    @property
    def loss_scale(self):
        return 1.0 / self.num_gpus        # detector.py:650-655

    def levels(self):
        return range(self.k_min, self.k_max + 1)


@dataclass
class HeadModel:
    """The model-helper surface the builders need (a cut-down
    DetectionModelHelper): a net, parameter bookkeeping, losses, metrics."""
    cfg: HeadConfig
    train: bool = True
    name: str = "retinanet_heads"
    net: core.Net = None
    params: list = field(default_factory=list)      # (name, shape, (filler, kwargs))
    weights: list = field(default_factory=list)
    biases: list = field(default_factory=list)
    losses: list = field(default_factory=list)
    metrics: list = field(default_factory=list)

    def __post_init__(self):
        if self.net is None:
            self.net = core.Net(self.name)

    def _conv_args(self):
        kw = dict(kernel=3, pad=1, stride=1, order="NCHW")
        if self.cfg.use_cudnn_engine:
            kw.update(engine="CUDNN", exhaustive_search=False)
        return kw

    def conv3x3(self, blob_in, blob_out, dim_in, dim_out, owner_prefix, weight_init, bias_init,
                relu=False):
        """Conv that creates `<blob_out>_w/_b` when it owns them, else reuses
        `<owner_prefix>_w/_b` (ConvShared, detector.py:449-482)."""
        w, b = owner_prefix + "_w", owner_prefix + "_b"
        if owner_prefix == blob_out:
            self.params.append((w, [dim_out, dim_in, 3, 3], weight_init))
            self.params.append((b, [dim_out], bias_init))
            self.weights.append(w)
            self.biases.append(b)
        kw = self._conv_args()
        if relu and self.cfg.fuse_relu:
            kw["fuse_relu"] = 1
        out = self.net.Conv([blob_in, w, b], blob_out, **kw)
        if relu and not self.cfg.fuse_relu:
            out = self.net.Relu(out, out)       # in place, retinanet_heads.py:124
        return out


def retinanet_bias_init(cfg):
    """cls_pred bias so that sigmoid(b) = prior_prob (retinanet_heads.py:29-60)."""
    return ("ConstantFill", {"value": float(-np.log((1 - cfg.prior_prob) / cfg.prior_prob))})


GAUSS = ("GaussianFill", {"std": 0.01})
ZERO = ("ConstantFill", {"value": 0.0})


def add_fpn_retinanet_outputs(model, blobs_in, dim_in, prefix=""):
    """blobs_in: FPN feature blobs ordered coarsest level first (k_max..k_min),
    as the FPN body returns them.  Returns {level: (cls_pred, bbox_pred)}."""
    cfg = model.cfg
    assert len(blobs_in) == cfg.k_max - cfg.k_min + 1
    A = cfg.num_anchors
    towers = (
        # (tower name, prediction blob stem, prediction width, prediction bias init)
        ("cls", "retnet_cls_pred", (cfg.num_classes - 1) * A, retinanet_bias_init(cfg)),
        ("bbox", "retnet_bbox_pred", 4 * A, ZERO),
    )
    feats = {}
    preds = {lvl: [None, None] for lvl in cfg.levels()}
    # The reference builds the whole cls subnet (towers + logits [+ teacher
    # sigmoid]) first, then the bbox towers, then the bbox predictions.
    for ti, (tower, pred_stem, pred_dim, pred_bias) in enumerate(towers):
        for lvl in cfg.levels():
            x = blobs_in[cfg.k_max - lvl]
            for i in range(cfg.num_convs):
                stem = prefix + "retnet_%s_conv_n%d_fpn" % (tower, i)
                x = model.conv3x3(x, stem + str(lvl), dim_in, dim_in, stem + str(cfg.k_min),
                                  GAUSS, ZERO, relu=True)
            feats[(tower, lvl)] = x
            if tower == "cls":
                preds[lvl][0] = _prediction(model, x, prefix + pred_stem, lvl, dim_in, pred_dim,
                                            pred_bias)
                if not model.train:
                    model.net.Sigmoid(preds[lvl][0], prefix + "retnet_cls_prob_fpn%d" % lvl)
        if tower == "bbox":
            for lvl in cfg.levels():
                preds[lvl][1] = _prediction(model, feats[(tower, lvl)], prefix + pred_stem, lvl,
                                            dim_in, pred_dim, pred_bias)
    return {lvl: tuple(p) for lvl, p in preds.items()}


def _prediction(model, feat, stem, lvl, dim_in, dim_out, bias_init):
    name = "%s_fpn%d" % (stem, lvl)
    owner = "%s_fpn%d" % (stem, model.cfg.k_min)
    return model.conv3x3(feat, name, dim_in, dim_out, owner, GAUSS, bias_init, relu=False)


def _loss_gradients(model, loss_blobs):
    """A gradient of 1 for each loss (utils/blob.py:166-172)."""
    return {str(b): str(model.net.ConstantFill(b, [str(b) + "_grad"], value=1.0))
            for b in loss_blobs}


def add_fpn_retinanet_losses(model):
    """The student's supervised losses (retinanet_heads.py:259-307).  Their
    operators (SelectSmoothL1Loss, SigmoidFocalLoss) are outside this build's
    hot path; the builder records them so the graph is complete."""
    cfg = model.cfg
    model.metrics += ["retnet_fg_num", "retnet_bg_num"]
    grads, losses = [], []
    for lvl in cfg.levels():
        s = "fpn%d" % lvl
        grads.append(model.net.SelectSmoothL1Loss(
            ["retnet_bbox_pred_" + s, "retnet_roi_bbox_targets_" + s,
             "retnet_roi_fg_bbox_locs_" + s, "retnet_fg_num"],
            "retnet_loss_bbox_" + s, beta=cfg.bbox_reg_beta,
            scale=cfg.loss_scale * cfg.bbox_reg_weight))
        losses.append("retnet_loss_bbox_" + s)
    for lvl in cfg.levels():
        s = "fpn%d" % lvl
        grads.append(model.net.SigmoidFocalLoss(
            ["retnet_cls_pred_" + s, "retnet_cls_labels_" + s, "retnet_fg_num"],
            ["fl_" + s], gamma=cfg.focal_gamma, alpha=cfg.focal_alpha, scale=cfg.loss_scale,
            num_classes=cfg.num_classes - 1))
        losses.append("fl_" + s)
    model.losses += losses
    return _loss_gradients(model, grads)


def add_distill_loss(model, student_prefix="", teacher_prefix="teacher/"):
    """PowSum normaliser over the teacher probabilities of all levels, then one
    SigmoidAdaptiveDistillLoss per level (retinanet_heads.py:313-352)."""
    cfg = model.cfg
    normalizer = student_prefix + "retnet_fg_num"
    if cfg.adaptive_normalizer:
        normalizer = student_prefix + "distill_normalizer"
        model.net.PowSum([teacher_prefix + "retnet_cls_prob_fpn%d" % l for l in cfg.levels()],
                         normalizer, power=cfg.logits_power)
        model.metrics.append(normalizer)
    grads, losses = [], []
    for lvl in cfg.levels():
        s = "fpn%d" % lvl
        out = student_prefix + "fl_distill_" + s
        grads.append(model.net.SigmoidAdaptiveDistillLoss(
            [student_prefix + "retnet_cls_pred_" + s, teacher_prefix + "retnet_cls_prob_" + s,
             "retnet_cls_labels_" + s, normalizer],
            [out], gamma=cfg.distill_gamma, alpha=cfg.distill_alpha,
            scale=cfg.loss_scale * cfg.temperature * cfg.temperature, beta=cfg.distill_beta,
            num_classes=cfg.num_classes - 1, ignored_label=cfg.ignored_label))
        losses.append(out)
    model.losses += losses
    return _loss_gradients(model, grads)
