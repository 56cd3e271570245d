"""Detectron weights files for the subnet parameters (utils/net.py)."""
import os
import pickle
from collections import OrderedDict

import numpy as np
import pytest
import torch

import ssad_amd  # noqa: F401
from ssad_amd.head_pipeline import FlatParams
from ssad_amd.modeling.retinanet_heads import HeadConfig
from ssad_amd.utils import net


class Store(object):
    def __init__(self, cfg):
        self.params = FlatParams(cfg, "cpu")
        self.moms = FlatParams(cfg, "cpu")
        self.teacher = FlatParams(cfg, "cpu")
        self.preserved = OrderedDict()


def _random_blobs(cfg, rng, prefix=""):
    blobs = {}
    for name, shape, _, _ in FlatParams(cfg, "cpu").specs:
        blobs[prefix + name] = rng.standard_normal(shape).astype(np.float32)
    return blobs


def test_round_trip_and_python2_layout(tmp_path):
    cfg = HeadConfig(num_convs=2)
    rng = np.random.default_rng(0)
    blobs = _random_blobs(cfg, rng)
    for k in list(blobs):
        blobs[k + "_momentum"] = rng.standard_normal(blobs[k].shape).astype(np.float32)
    blobs["conv1_w"] = rng.standard_normal((64, 3, 7, 7)).astype(np.float32)      # backbone blob
    blobs.update(_random_blobs(cfg, rng, "teacher/"))
    path = str(tmp_path / "model_final.pkl")
    with open(path, "wb") as f:       # what the Python 2 reference writes
        pickle.dump(dict(blobs=blobs, cfg="NUM_GPUS: 8\n"), f, protocol=2)

    st = Store(cfg)
    loaded, missing = net.initialize_from_weights_file(st, path)
    assert not missing
    for name, _, _, _ in st.params.specs:
        assert np.array_equal(st.params[name].numpy(), blobs[name])
        assert np.array_equal(st.moms[name].numpy(), blobs[name + "_momentum"])
        assert np.array_equal(st.teacher[name].numpy(), blobs["teacher/" + name])
    assert list(st.preserved) == ["conv1_w"]

    out = str(tmp_path / "model_iter9.pkl")
    net.save_model_to_weights_file(out, st, cfg_yaml="NUM_GPUS: 1\n")
    saved = pickle.load(open(out, "rb"))
    assert set(saved) == {"blobs", "cfg"} and saved["cfg"] == "NUM_GPUS: 1\n"
    for name, _, _, _ in st.params.specs:
        assert np.array_equal(saved["blobs"][name], blobs[name])
        assert np.array_equal(saved["blobs"][name + "_momentum"], blobs[name + "_momentum"])
    assert np.array_equal(saved["blobs"]["conv1_w"], blobs["conv1_w"])
    # the teacher/ scope is saved with the model (net.py:145-152 walks model.params, which holds
    # the teacher's parameters): the file alone resumes distillation
    for name, _, _, _ in st.teacher.specs:
        assert np.array_equal(saved["blobs"]["teacher/" + name], blobs["teacher/" + name])
    st2 = Store(cfg)
    _, missing2 = net.initialize_from_weights_file(st2, out)
    assert not missing2 and torch.equal(st2.teacher.flat, st.teacher.flat)


def test_bare_dict_missing_and_mismatched_blobs(tmp_path):
    cfg = HeadConfig(num_convs=1)
    rng = np.random.default_rng(1)
    blobs = _random_blobs(cfg, rng)
    dropped = "retnet_bbox_pred_fpn3_b"
    del blobs[dropped]
    bad = "retnet_cls_pred_fpn3_w"
    blobs[bad] = np.zeros((81 * 9, 256, 3, 3), np.float32)      # softmax-style head: other shape
    path = str(tmp_path / "old_style.pkl")
    pickle.dump(blobs, open(path, "wb"), protocol=2)             # pre-'blobs' layout
    tpath = str(tmp_path / "teacher.pkl")
    tblobs = _random_blobs(cfg, rng)
    pickle.dump(dict(blobs=tblobs), open(tpath, "wb"), protocol=2)

    st = Store(cfg)
    before = st.params[bad].clone()
    loaded, missing = net.initialize_from_weights_file(st, path, teacher_weights_file=tpath)
    assert missing == [dropped]
    assert bad not in loaded and torch.equal(st.params[bad], before)     # skipped, untouched
    for name in tblobs:
        assert np.array_equal(st.teacher[name].numpy(), tblobs[name])
    assert float(st.moms.flat.abs().sum()) == 0.0                        # no history in the file


def test_mismatched_blob_skips_its_momentum_too(tmp_path):
    """net.py:106-109: on a shape mismatch the reference `continue`s before it feeds either the
    parameter or `<name>_momentum`."""
    cfg = HeadConfig(num_convs=1)
    rng = np.random.default_rng(2)
    blobs = _random_blobs(cfg, rng)
    bad = "retnet_cls_pred_fpn3_b"
    for k in list(blobs):
        blobs[k + "_momentum"] = rng.standard_normal(blobs[k].shape).astype(np.float32)
    blobs[bad] = np.zeros(7, np.float32)
    blobs[bad + "_momentum"] = np.ones(blobs[bad + "_momentum"].shape, np.float32)   # right shape
    path = str(tmp_path / "m.pkl")
    pickle.dump(dict(blobs=blobs), open(path, "wb"), protocol=2)
    st = Store(cfg)
    loaded, _ = net.initialize_from_weights_file(st, path)
    assert bad not in loaded
    assert float(st.moms[bad].abs().sum()) == 0.0            # its momentum was not fed
    other = "retnet_bbox_pred_fpn3_b"
    assert np.array_equal(st.moms[other].numpy(), blobs[other + "_momentum"])
    assert bad + "_momentum" not in st.preserved


@pytest.mark.gpu
def test_update_lr_rescales_update_history_and_checkpoint_round_trip(tmp_path):
    from ssad_amd.head_pipeline import DistillHeads
    cfg = HeadConfig(num_convs=1)
    heads = DistillHeads(cfg=cfg, N=1, shapes=[(8, 8), (4, 4)], lr=0.01)
    heads.moms.flat.fill_(2.0)
    # warm-up sized change (ratio 1.05 <= SCALE_MOMENTUM_THRESHOLD): lr only
    heads.update_lr(0.0105)
    assert float(heads.lr) == float(np.float32(0.0105))
    assert float(heads.moms.flat.min()) == 2.0 and float(heads.moms.flat.max()) == 2.0
    # step decay x0.1: V is rescaled by new/old (detector.py:616-648)
    heads.update_lr(0.00105)
    want = np.float32(2.0) * np.float32(np.float32(0.00105) / np.float32(0.0105))
    assert torch.allclose(heads.moms.flat, torch.full_like(heads.moms.flat, float(want)), rtol=1e-6)
    # from (near) zero: no correction (cur_lr > 1e-7 guard)
    heads.lr.fill_(0.0)
    heads.update_lr(0.01)
    assert torch.allclose(heads.moms.flat, torch.full_like(heads.moms.flat, float(want)), rtol=1e-6)

    torch.manual_seed(0)
    heads.params.flat.normal_()
    path = str(tmp_path / "ckpt.pkl")
    net.save_model_to_weights_file(path, heads)
    other = DistillHeads(cfg=cfg, N=1, shapes=[(8, 8), (4, 4)], lr=0.01)
    loaded, missing = net.initialize_from_weights_file(other, path)
    assert not missing and len(loaded) == 2 * len(heads.params.specs)      # student + teacher/ scope
    assert torch.equal(other.teacher.flat, heads.teacher.flat)
    assert torch.equal(other.params.flat, heads.params.flat)
    assert torch.equal(other.moms.flat, heads.moms.flat)
