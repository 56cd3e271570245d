"""Detectron weight files for the RetinaNet subnets (row f3).

Mirror of detectron/lib/utils/net.py:40-182 for the parameters this repo owns:
a weights file is a pickled dict `{'blobs': {name: ndarray}, 'cfg': yaml}` (or,
older, the blob dict itself) with parameters under their unscoped Caffe2 names
(`retnet_cls_conv_n0_fpn3_w`, ...), update history under `<name>_momentum`
(net.py:101-104,136-143) and everything the current model does not use carried
along untouched (the reference parks those under `__preserve__/`, net.py:120-133).
Files written by the Python 2 reference load here (latin-1 unpickling), files
written here use pickle protocol 2 so the reference can read them.

    store = heads (a DistillHeads, or anything with .params / .moms / .teacher
            FlatParams and a .preserved dict)
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
    student file carries under `teacher/` (net.py:71-78).  Returns the lists of
    loaded and missing parameter names."""
    logger.info("Loading weights from: %s", weights_file)
    src, _ = _blobs_and_cfg(load_object(weights_file))
    src = dict(src)
    if teacher_weights_file is not None:
        logger.info("Loading teacher weights from: %s", teacher_weights_file)
        tsrc, _ = _blobs_and_cfg(load_object(teacher_weights_file))
        for k, v in tsrc.items():
            src["teacher/" + k] = v
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
            if teacher_weights_file is not None or any(k.startswith("teacher/") for k in src):
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
