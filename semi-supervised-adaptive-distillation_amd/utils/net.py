"""Detectron weight files for the RetinaNet subnets and the native ResNet-FPN backbones (row f3).

Mirror of detectron/lib/utils/net.py:40-182 for the parameters this repo owns:
a weights file is a pickled dict `{'blobs': {name: ndarray}, 'cfg': yaml}` (or,
older, the blob dict itself) with parameters under their unscoped Caffe2 names
(`retnet_cls_conv_n0_fpn3_w`, ...), update history under `<name>_momentum`
(net.py:101-104,136-143) and everything the current model does not use carried
along untouched (the reference parks those under `__preserve__/`, net.py:120-133).
Files written by the Python 2 reference load here (latin-1 unpickling), files
written here use pickle protocol 2 so the reference can read them.

    store = heads (a DistillHeads, or anything with .params / .moms / .teacher
            FlatParams and a .preserved dict), or a whole detector
            (backbone_pipeline.NativeDistillModel: .heads + .student [+ .teacher] backbones)

Backbones.  The reference keeps a body convolution and the frozen BatchNorm that follows it as
three blobs -- `res4_2_branch2b_w`, `res4_2_branch2b_bn_s`, `res4_2_branch2b_bn_b`
(detectron/lib/modeling/ResNet.py:221-283: Conv with no_bias, then AffineChannel) -- while
backbone_pipeline.NativeResNetFPN stores the pair folded, W' = s W per output channel and bias = b.
Loading folds, saving un-folds (W = W' / s; the scale and bias blobs are written back as they were
read, they are never trained: affine_channel_op.cc registers a gradient for X only).  The update
history of a folded filter relates the same way: `<w>_momentum` = m, m' = s m.  Blob names follow
ResNet.py / FPN.py (captured from the imported reference builder in
tests/golden/backbone_graph_*.json): conv1_w + res_conv1_bn_*, res{2..5}_{i}_branch2{a,b,c}_*,
res{N}_{0}_branch1_* (projection), fpn_inner_res5_{n}_sum_*, fpn_inner_res{4,3}_{n}_sum_lateral_*,
fpn_res{5,4,3}_{n}_sum_*, fpn_6_*, fpn_7_*; the teacher's under `teacher/` (net.py:71-78).
"""
import logging
import os
import pickle
from collections import OrderedDict

import numpy as np
import torch

logger = logging.getLogger(__name__)


def load_object(path):
    with open(path, "rb") as f:
        try:
            return pickle.load(f)
        except UnicodeDecodeError:
            f.seek(0)
            return pickle.load(f, encoding="latin1")      # written by Python 2


def _blobs_and_cfg(obj):
    cfg_yaml = None
    if isinstance(obj, dict) and "cfg" in obj:
        cfg_yaml = obj["cfg"]
    if isinstance(obj, dict) and "blobs" in obj:          # net.py:66-69 backwards compat
        obj = obj["blobs"]
    return obj, cfg_yaml


def _feed(flat, name, arr):
    """FeedBlob with the reference's shape check (net.py:106-119): a mismatching
    source blob is reported and skipped."""
    dst = flat[name]
    arr = np.asarray(arr)
    if tuple(dst.shape) != tuple(arr.shape):
        logger.info("Shape missmatch: name: %s src: %s, dst: %s", name, tuple(dst.shape),
                    tuple(arr.shape))
        return False
    dst.copy_(torch.as_tensor(arr.astype(np.float32, copy=False)))
    return True


def initialize_from_weights_file(store, weights_file, teacher_weights_file=None):
    """Load student parameters (+ momentum) and, for distillation training, the
    teacher's parameters -- either from `teacher_weights_file` or from blobs the
    student file carries under `teacher/` (net.py:71-78).  `store`: the subnets
    (DistillHeads / FlatParams holder) or a whole NativeDistillModel, whose backbones
    are then fed too.  Returns the lists of loaded and missing parameter names."""
    logger.info("Loading weights from: %s", weights_file)
    src, _ = _blobs_and_cfg(load_object(weights_file))
    src = dict(src)
    if teacher_weights_file is not None:
        logger.info("Loading teacher weights from: %s", teacher_weights_file)
        tsrc, _ = _blobs_and_cfg(load_object(teacher_weights_file))
        for k, v in tsrc.items():
            src["teacher/" + k] = v
    if hasattr(store, "student") and hasattr(store, "heads"):
        return initialize_from_blobs(store, src)
    return _initialize_heads(store, src, teacher_weights_file is not None)


def _initialize_heads(store, src, teacher_file_given=False):
    loaded, missing = [], []
    used = set()
    for name, _, _, _ in store.params.specs:
        if name not in src:
            logger.info("%s not found", name)
            missing.append(name)
            continue
        used.add(name)
        mname = name + "_momentum"
        if mname in src:
            used.add(mname)
        if not _feed(store.params, name, src[name]):
            continue            # net.py:106-109: a mismatching blob is skipped, momentum included
        loaded.append(name)
        if mname in src:
            _feed(store.moms, name, src[mname])
    for name, _, _, _ in store.teacher.specs:
        tname = "teacher/" + name
        if tname not in src:
            if teacher_file_given or any(k.startswith("teacher/") for k in src):
                logger.info("%s not found", tname)
                missing.append(tname)
            continue
        used.add(tname)
        if _feed(store.teacher, name, src[tname]):
            loaded.append(tname)
    # blobs the subnets do not own (backbone, other heads) ride along to the next save
    preserved = getattr(store, "preserved", None)
    if preserved is None:
        preserved = store.preserved = OrderedDict()
    for k, v in src.items():
        if k not in used and not k.endswith("_momentum") and v is not None and \
                not k.startswith("teacher/"):
            preserved[k] = v
    return loaded, missing


def save_model_to_weights_file(weights_file, store, cfg_yaml=""):
    """Parameters, their update history and the preserved blobs, as
    save_model_to_weights_file (net.py:137-168) writes them."""
    logger.info("Saving parameters and momentum to %s", os.path.abspath(weights_file))
    blobs = OrderedDict()
    model = None
    if hasattr(store, "student") and hasattr(store, "heads"):        # the whole detector
        model, store = store, store.heads
        blobs.update(backbone_to_blobs(model.student))
        if model.teacher is not None:
            blobs.update(backbone_to_blobs(model.teacher, "teacher/", momentum=False))
    for name, _, _, _ in store.params.specs:
        blobs[name] = store.params[name].detach().cpu().numpy().copy()
    for name, _, _, _ in store.params.specs:
        blobs[name + "_momentum"] = store.moms[name].detach().cpu().numpy().copy()
    # the reference saves every blob of model.params, the teacher/ scope included (net.py:145-152
    # walks model.params, which holds the teacher's parameters in a distillation model), so a
    # file written here resumes distillation without the separate teacher file
    teacher = getattr(store, "teacher", None)
    if teacher is not None and getattr(store, "distill", True):
        for name, _, _, _ in teacher.specs:
            blobs["teacher/" + name] = teacher[name].detach().cpu().numpy().copy()
    for k, v in getattr(store, "preserved", {}).items():
        if k not in blobs:
            blobs[k] = v
    tmp = weights_file + ".tmp"
    with open(tmp, "wb") as f:
        pickle.dump(dict(blobs=dict(blobs), cfg=cfg_yaml), f, protocol=2)
    os.replace(tmp, weights_file)


# ---------------------------------------------------------------------------
# ResNet / ResNeXt-FPN backbones
# ---------------------------------------------------------------------------

_BLOCKS = {"r50": (3, 4, 6, 3), "r101": (3, 4, 23, 3), "x101-64x4d": (3, 4, 23, 3)}


def backbone_blob_names(arch):
    """Native layer name -> (filter blob, AffineChannel scale blob or None, bias blob): the
    reference's names for every convolution of a ResNet / ResNeXt-FPN RetinaNet body
    (ResNet.py:85-130,221-283 add_stage / bottleneck_transformation / basic_bn_shortcut /
    basic_bn_stem; FPN.py:116-250 with the RetinaNet levels P3..P7).  Body layers have a scale
    blob and their "bias" is the AffineChannel's; the FPN's own convolutions have a real bias."""
    if arch not in _BLOCKS:
        raise ValueError("backbone_blob_names: architectures %s" % sorted(_BLOCKS))
    names = OrderedDict()
    names["stem.0"] = ("conv1_w", "res_conv1_bn_s", "res_conv1_bn_b")
    blocks = _BLOCKS[arch]
    for si, n in enumerate(blocks):
        stage = si + 2
        for j in range(n):
            pre = "res%d_%d" % (stage, j)
            for part, br in (("c1", "branch2a"), ("c2", "branch2b"), ("c3", "branch2c")):
                b = "%s_%s" % (pre, br)
                names["res%d.%d.%s" % (stage, j, part)] = (b + "_w", b + "_bn_s", b + "_bn_b")
            if j == 0:          # the stage's first block changes width (and stride): projection shortcut
                b = pre + "_branch1"
                names["res%d.%d.proj" % (stage, j)] = (b + "_w", b + "_bn_s", b + "_bn_b")
    last = {stage: "res%d_%d_sum" % (stage, blocks[stage - 2] - 1) for stage in (3, 4, 5)}
    for i, stage in enumerate((5, 4, 3)):
        inner = "fpn_inner_%s" % last[stage] + ("" if stage == 5 else "_lateral")
        names["lat.%d" % i] = (inner + "_w", None, inner + "_b")
        names["out.%d" % i] = ("fpn_%s_w" % last[stage], None, "fpn_%s_b" % last[stage])
    names["p6"] = ("fpn_6_w", None, "fpn_6_b")
    names["p7"] = ("fpn_7_w", None, "fpn_7_b")
    return names


def backbone_from_blobs(blobs, arch, prefix="", strict=True):
    """Fold a Detectron blob dict into what NativeResNetFPN(src=, affine_scales=) takes.
    -> (state {layer.weight / layer.bias: float32 tensor}, scales {layer: [cout] float32 tensor},
        momentum {layer.weight / layer.bias: tensor} for the blobs that carry `_momentum`,
        missing [blob names])
    strict: a missing blob raises KeyError (a backbone with holes computes garbage); otherwise the
    layer is left out of `state` and reported."""
    state, scales, moms, missing = {}, {}, {}, []
    for layer, (wn, sn, bn) in backbone_blob_names(arch).items():
        need = [n for n in (wn, sn, bn) if n is not None]
        absent = [prefix + n for n in need if prefix + n not in blobs]
        if absent:
            if strict:
                raise KeyError("weights file lacks %s (layer %s of %s)" % (", ".join(absent), layer, arch))
            missing += absent
            continue
        w = torch.as_tensor(np.asarray(blobs[prefix + wn], dtype=np.float32))
        b = torch.as_tensor(np.asarray(blobs[prefix + bn], dtype=np.float32)).reshape(-1)
        if sn is not None:
            sc = torch.as_tensor(np.asarray(blobs[prefix + sn], dtype=np.float32)).reshape(-1)
            if sc.numel() != w.shape[0] or b.numel() != w.shape[0]:
                msg = "%s: AffineChannel blobs of %d / %d channels for a filter with %d outputs" % (
                    prefix + wn, sc.numel(), b.numel(), w.shape[0])
                if strict:
                    raise ValueError(msg)
                logger.info("Shape missmatch: %s", msg)      # net.py:106-119: reported and skipped
                missing += [prefix + n for n in need]
                continue
            scales[layer] = sc
            w = w * sc.view(-1, 1, 1, 1)
        state[layer + ".weight"], state[layer + ".bias"] = w, b
        m = blobs.get(prefix + wn + "_momentum")
        if m is not None:
            m = torch.as_tensor(np.asarray(m, dtype=np.float32))
            moms[layer + ".weight"] = m * scales[layer].view(-1, 1, 1, 1) if sn is not None else m
        if sn is None and blobs.get(prefix + bn + "_momentum") is not None:
            moms[layer + ".bias"] = torch.as_tensor(np.asarray(blobs[prefix + bn + "_momentum"], dtype=np.float32))
    return state, scales, moms, missing


def load_backbone(net, blobs, prefix="", load_momentum=True, strict=False):
    """Feed an existing NativeResNetFPN / NativeResNetFPNF16 from a Detectron blob dict (the
    network must have been built with per-layer scale slots: src= / affine_scales= given, which
    the constructors of this module's callers do).  Remembers the scale blobs for saving.

    strict=False is the reference's behaviour (net.py:96-99, :106-119): a layer whose blobs the file
    does not hold is logged as "<name> not found" and keeps its initialised filter, bias and scale slot;
    a blob of another shape is logged and skipped the same way.  The names are left in
    `net.missing_blobs`.  strict=True raises KeyError / ValueError instead (a body with holes computes
    garbage unless the caller meant it: the reference's standard TRAIN.WEIGHTS is an ImageNet body that
    holds no fpn_* blob)."""
    state, scales, moms, missing = backbone_from_blobs(blobs, net.arch, prefix, strict=strict)
    for n in missing:
        logger.info("%s not found", n)
    names = backbone_blob_names(net.arch)
    for name, layer in net._layers.items():
        if name + ".weight" not in state:
            continue
        want = tuple(state[name + ".weight"].shape)
        have = (layer.cout, layer.wcin, layer.k, layer.k)
        if want != have:
            if strict:
                raise ValueError("%s: filter blob %s of shape %s, the %s network has %s" % (
                    name, names[name][0], want, net.arch, have))
            logger.info("Shape missmatch: name: %s src: %s, dst: %s", prefix + names[name][0], want, have)
            missing += [prefix + n for n in names[name] if n is not None]
            for d in (state, moms):
                d.pop(name + ".weight", None)
                d.pop(name + ".bias", None)
            scales.pop(name, None)
    net.missing_blobs = list(missing)
    net.load_from(state, affine_scales=scales, strict=strict)
    if load_momentum and net.train and moms:
        for name, layer in net._layers.items():
            if not layer.train:
                continue
            off = (layer.w.data_ptr() - net.params_flat.data_ptr()) // 4
            if name + ".weight" in moms:
                net.moms_flat[off:off + layer.w.numel()].copy_(moms[name + ".weight"].reshape(-1).to(net.device))
            if layer.gb is not None and name + ".bias" in moms:
                ob = (layer.b.data_ptr() - net.params_flat.data_ptr()) // 4
                net.moms_flat[ob:ob + layer.cout].copy_(moms[name + ".bias"].to(net.device))
    return net


def backbone_to_blobs(net, prefix="", momentum=True):
    """The network's parameters in the reference's layout: filters un-folded (W = W' / s), the
    AffineChannel blobs as loaded (or the construction-time scales and the folded biases),
    `_momentum` for what the reference trains (model.TrainableParams(): the filters of res3..res5
    and the FPN's filters and biases; net.py:137-168)."""
    out = OrderedDict()
    aff = getattr(net, "affine_scale_values", {})
    for name, (wn, sn, bn) in backbone_blob_names(net.arch).items():
        layer = net._layers[name]
        w = layer.w.detach().float().cpu()
        b = layer.b.detach().float().cpu()
        m = mb = None
        if momentum and layer.train:
            off = (layer.w.data_ptr() - net.params_flat.data_ptr()) // 4
            m = net.moms_flat[off:off + layer.w.numel()].detach().cpu().view_as(w)
            if layer.gb is not None:
                ob = (layer.b.data_ptr() - net.params_flat.data_ptr()) // 4
                mb = net.moms_flat[ob:ob + layer.cout].detach().cpu()
        if sn is not None:
            sc = aff.get(name)
            sc = torch.ones(layer.cout) if sc is None else torch.as_tensor(sc, dtype=torch.float32).reshape(-1).cpu()
            if sc.numel() == 1:
                sc = sc.expand(layer.cout).contiguous()
            safe = torch.where(sc != 0, sc, torch.ones_like(sc)).view(-1, 1, 1, 1)
            live = (sc != 0).view(-1, 1, 1, 1)
            w = torch.where(live, w / safe, torch.zeros_like(w))       # a dead channel (s = 0) has no W to recover
            if m is not None:
                m = torch.where(live, m / safe, torch.zeros_like(m))
            out[prefix + sn] = sc.numpy().copy()
        out[prefix + wn] = w.numpy().copy()
        out[prefix + bn] = b.numpy().copy()
        if m is not None:
            out[prefix + wn + "_momentum"] = m.numpy().copy()
        if mb is not None:
            out[prefix + bn + "_momentum"] = mb.numpy().copy()
    return out


def native_model_from_weights_files(heads, weights_file, teacher_weights_file=None, student_arch="r50",
                                    teacher_arch="r101", strict=False, **model_kw):
    """initialize_from_weights_file (net.py:50-147) for the whole detector: builds a
    backbone_pipeline.NativeDistillModel whose backbones AND subnets hold the weights of
    `weights_file` (student; its `teacher/` blobs or `teacher_weights_file` for the teacher:
    net.py:71-78), momentum included.  Blobs the file lacks keep the model's initialisation and are
    returned in `missing`, as the reference does (an ImageNet body-only R-50.pkl is the standard
    TRAIN.WEIGHTS); strict=True raises on the first hole.  -> (model, loaded names, missing names)"""
    from ..backbone_pipeline import NativeDistillModel
    src, _ = _blobs_and_cfg(load_object(weights_file))
    src = dict(src)
    if teacher_weights_file is not None:
        tsrc, _ = _blobs_and_cfg(load_object(teacher_weights_file))
        for k, v in tsrc.items():
            src["teacher/" + k] = v
    s_state, s_scales, _, _ = backbone_from_blobs(src, student_arch, strict=strict)
    has_teacher = teacher_arch not in (None, "none")
    t_state = t_scales = None
    if has_teacher:
        t_state, t_scales, _, _ = backbone_from_blobs(src, teacher_arch, "teacher/", strict=strict)
    model = NativeDistillModel(heads, student_arch, teacher_arch if has_teacher else None, student_src=s_state,
                               teacher_src=t_state, student_scales=s_scales, **model_kw)
    loaded, missing = initialize_from_blobs(model, src, strict=strict)
    return model, loaded, missing


def initialize_from_blobs(model, src, strict=False):
    """Feed a built NativeDistillModel (subnets + both backbones) from a blob dict that already
    carries the teacher under `teacher/`.  strict: see load_backbone."""
    loaded, missing = _initialize_heads(model.heads, src)
    nets = [(model.student, "")] + ([(model.teacher, "teacher/")] if model.teacher is not None else [])
    for net, prefix in nets:
        load_backbone(net, src, prefix, load_momentum=not prefix, strict=strict)
        gone = set(net.missing_blobs)
        missing += net.missing_blobs
        loaded += [prefix + n for t in backbone_blob_names(net.arch).values() for n in t
                   if n is not None and prefix + n not in gone]
    owned = set(loaded) | set(n + "_momentum" for n in loaded)
    for k in list(model.heads.preserved):
        if k in owned:
            del model.heads.preserved[k]          # the backbones own these now
    # rank 0's loaded state is what every replica starts from (net.py:185-208 broadcast_parameters walks all of
    # model.params): trained parameters and history, the frozen values and s^2 slots of the student's backbone,
    # the whole frozen teacher -- subnets and backbone
    model.heads.broadcast_params()
    model.student.broadcast_params()
    if model.teacher is not None:
        model.teacher.broadcast_params()
    return loaded, missing
