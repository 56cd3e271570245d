"""CPU restatement of one distillation iteration of the RetinaNet subnets,
composed from the oracle's operators in the order of the reference graph
(detectron/lib/modeling/retinanet_heads.py:63-245,313-352 forward;
caffe2 autograd backward incl. the per-level gradient Sum of shared weights,
caffe2/python/core.py:706-741).  TEST INFRASTRUCTURE / cpu_baseline only."""
import numpy as np

from . import oracle


def _layers(tower, num_convs=4, k_min=3):
    names = ["retnet_%s_conv_n%d_fpn%d" % (tower, i, k_min) for i in range(num_convs)]
    return names + ["retnet_%s_pred_fpn%d" % (tower, k_min)]


def tower_forward(params, tower, feats, keep=None):
    """feats: list of per-level arrays.  Returns per-level prediction arrays;
    keep (optional list) receives the post-ReLU activations per layer."""
    layers = _layers(tower)
    xs = feats
    for name in layers[:-1]:
        xs = [oracle.relu(oracle.conv_forward(x, params[name + "_w"], params[name + "_b"]))
              for x in xs]
        if keep is not None:
            keep.append(xs)
    name = layers[-1]
    return [oracle.conv_forward(x, params[name + "_w"], params[name + "_b"]) for x in xs]


def tower_backward(params, tower, feats, acts, d_pred):
    """Returns ({param: grad}, d_feats)."""
    layers = _layers(tower)
    grads = {}
    dy = d_pred
    for li in range(len(layers) - 1, -1, -1):
        name = layers[li]
        x_in = acts[li - 1] if li > 0 else feats
        gw = np.zeros_like(params[name + "_w"])
        gb = np.zeros_like(params[name + "_b"])
        dxs = []
        for x, d in zip(x_in, dy):          # autograd Sum over the levels
            w_, b_, x_ = oracle.conv_backward(x, params[name + "_w"], d)
            gw += w_
            gb += b_
            dxs.append(oracle.relu_grad(x, x_) if li > 0 else x_)
        grads[name + "_w"], grads[name + "_b"] = gw, gb
        dy = dxs
    return grads, dy


def head_step(student, teacher, fpn_student, fpn_teacher, labels, d_bbox_pred=None, *,
              num_classes=80, gamma=2.0, alpha=0.5, beta=0.0, ignored_label=-1, scale=1.0,
              power=1.8, teacher_bbox_tower=True, bbox_targets=None, fg_num=None,
              focal_gamma=2.0, focal_alpha=0.25, bbox_beta=0.11, loss_scale=None):
    """With bbox_targets / fg_num the student's supervised losses
    (SigmoidFocalLoss + SelectSmoothL1Loss, retinanet_heads.py:259-307) join
    the distillation loss, as in the reference graph."""
    loss_scale = scale if loss_scale is None else loss_scale
    distill = teacher is not None
    t_prob, norm64 = None, None
    if distill:
        t_logits = tower_forward(teacher, "cls", fpn_teacher)
        t_prob = [oracle.sigmoid(x) for x in t_logits]
        if teacher_bbox_tower:
            tower_forward(teacher, "bbox", fpn_teacher)
    acts = {"cls": [], "bbox": []}
    cls_logits = tower_forward(student, "cls", fpn_student, acts["cls"])
    bbox_pred = tower_forward(student, "bbox", fpn_student, acts["bbox"])
    losses, d_logits = [], []
    if distill:
        norm32, norm64 = oracle.pow_sum(t_prob, power)
        kw = dict(gamma=gamma, alpha=alpha, beta=beta, num_classes=num_classes,
                  ignored_label=ignored_label, scale=scale)
        for x, q, g in zip(cls_logits, t_prob, labels):
            _, l64, _ = oracle.distill_loss_forward(x, q, g, norm32, **kw)
            losses.append(l64)
            d_logits.append(oracle.distill_loss_backward(x, q, g, norm32, 1.0, **kw))
    else:
        # plain RetinaNet (model_builder.py:98-100,413: `retinanet` without the distillation
        # wrapper; BASELINE config 2): only the supervised losses drive the subnets
        assert bbox_targets is not None, "student-only training needs the supervised losses"
        d_logits = [np.zeros_like(x) for x in cls_logits]
    focal_losses, bbox_losses = [], []
    if bbox_targets is not None:
        fkw = dict(gamma=focal_gamma, alpha=focal_alpha, num_classes=num_classes, scale=loss_scale)
        d_bbox_pred = []
        for i, (x, g) in enumerate(zip(cls_logits, labels)):
            focal_losses.append(oracle.focal_loss_forward(x, g, fg_num, **fkw)[1])
            d_logits[i] = d_logits[i] + oracle.focal_loss_backward(x, g, fg_num, 1.0, **fkw)
        for pred, (Y, Lc) in zip(bbox_pred, bbox_targets):
            bbox_losses.append(oracle.select_smooth_l1_forward(pred, Y, Lc, fg_num, beta=bbox_beta,
                                                               scale=loss_scale)[1])
            d_bbox_pred.append(oracle.select_smooth_l1_backward(pred, Y, Lc, fg_num, 1.0,
                                                                beta=bbox_beta, scale=loss_scale))
    grads, d_fpn = {}, {}
    g, d_fpn["cls"] = tower_backward(student, "cls", fpn_student, acts["cls"], d_logits)
    grads.update(g)
    g, d_fpn["bbox"] = tower_backward(student, "bbox", fpn_student, acts["bbox"], d_bbox_pred)
    grads.update(g)
    return dict(losses=np.array(losses), focal_losses=np.array(focal_losses),
                bbox_losses=np.array(bbox_losses), normalizer=norm64, grads=grads, d_fpn=d_fpn,
                cls_logits=cls_logits, bbox_pred=bbox_pred, t_prob=t_prob, d_logits=d_logits)
