"""Seeded synthetic inputs for the hot path.

COCO-shaped: a 600 px image is padded to 3x640x896, so the FPN levels P3..P7
are 80x112, 40x56, 20x28, 10x14, 5x7; A = 9 anchors, C = 80 classes.
"""
import numpy as np

LEVEL_SHAPES_600 = [(80, 112), (40, 56), (20, 28), (10, 14), (5, 7)]
LEVEL_SHAPES_500 = [(64, 96), (32, 48), (16, 24), (8, 12), (4, 6)]
NUM_ANCHORS = 9
NUM_CLASSES = 80
FPN_DIM = 256


def sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(np.float32)


def distill_inputs(rng, N, A, C, H, W, clip=1e-6):
    """logits ~ N(-4, 2^2); teacher prob = sigmoid(N(-4, 2^2)) clipped to
    [clip, 1-clip]; labels int32: 5 % -1 (ignore), 2 % uniform{1..C}, rest 0."""
    logits = (rng.standard_normal((N, A * C, H, W)) * 2.0 - 4.0).astype(np.float32)
    t_logit = (rng.standard_normal((N, A * C, H, W)) * 2.0 - 4.0).astype(np.float32)
    teacher = np.clip(sigmoid(t_logit), clip, 1.0 - clip).astype(np.float32)
    u = rng.random((N, A, H, W))
    labels = np.zeros((N, A, H, W), np.int32)
    labels[u < 0.05] = -1
    fg = (u >= 0.05) & (u < 0.07)
    labels[fg] = rng.integers(1, C + 1, size=int(fg.sum()), dtype=np.int32)
    return logits, teacher, labels


def fpn_features(rng, N, shapes=LEVEL_SHAPES_600, dim=FPN_DIM):
    return [rng.standard_normal((N, dim, h, w)).astype(np.float32)
            for (h, w) in shapes]


def head_params(rng, dim=FPN_DIM, A=NUM_ANCHORS, C=NUM_CLASSES, num_convs=4,
                prior_prob=0.01):
    """RetinaNet head parameters with the reference initialisation
    (detectron/lib/modeling/retinanet_heads.py:29-60,97-152): W ~ N(0, 0.01^2),
    b = 0, cls_pred bias = -log((1-pi)/pi)."""
    p = {}
    for tower in ("cls", "bbox"):
        for i in range(num_convs):
            p["retnet_%s_conv_n%d_fpn3_w" % (tower, i)] = (
                rng.standard_normal((dim, dim, 3, 3)) * 0.01).astype(np.float32)
            p["retnet_%s_conv_n%d_fpn3_b" % (tower, i)] = np.zeros(dim, np.float32)
    p["retnet_cls_pred_fpn3_w"] = (
        rng.standard_normal((A * C, dim, 3, 3)) * 0.01).astype(np.float32)
    p["retnet_cls_pred_fpn3_b"] = np.full(
        A * C, -np.log((1.0 - prior_prob) / prior_prob), np.float32)
    p["retnet_bbox_pred_fpn3_w"] = (
        rng.standard_normal((4 * A, dim, 3, 3)) * 0.01).astype(np.float32)
    p["retnet_bbox_pred_fpn3_b"] = np.zeros(4 * A, np.float32)
    return p


def bbox_targets(rng, labels, bbox_dim=36):
    """Foreground box-regression inputs of SelectSmoothL1Loss for one level:
    for every foreground anchor (label > 0) a row (n, 4*a, y, x) in L and a
    4-vector target in Y (detectron/lib/roi_data/retinanet.py:262-306 builds
    them the same way).  Returns (Y [M,4] f32, L [M,4] f32)."""
    n, a, y, x = np.nonzero(labels > 0)
    L = np.stack([n, 4 * a, y, x], axis=1).astype(np.float32)
    Y = (rng.standard_normal((L.shape[0], 4)) * 0.5).astype(np.float32)
    return Y, L
