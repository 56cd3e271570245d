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


# ---------------------------------------------------------------------------
# backbones: the reference's blob layout <-> NativeResNetFPN's folded layout
# ---------------------------------------------------------------------------

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GRAPHS = {"r50": "backbone_graph_r50_fpn.json", "x101-64x4d": "backbone_graph_x101_64x4d_fpn.json"}


def reference_backbone_blobs(arch, rng, prefix="", momentum=True):
    """A synthetic weights-file body for `arch` in the REFERENCE's layout: every parameter blob the
    imported reference builder creates (names and shapes from tests/golden/backbone_graph_*.json,
    captured by tests/golden/make_backbone_graph.py), random values; AffineChannel scales of both
    signs away from zero; `_momentum` for what the reference trains (filters of res3.., FPN)."""
    import json
    g = json.load(open(os.path.join(GOLDEN, GRAPHS[arch])))
    blobs = OrderedDict()
    for prm in g["params"]:
        name, shape = prm["name"], tuple(prm["shape"])
        if name.endswith("_bn_s"):
            v = rng.uniform(0.5, 1.5, shape) * rng.choice([-1.0, 1.0], shape, p=[0.1, 0.9])
        elif name.endswith("_bn_b") or name.endswith("_b"):
            v = rng.standard_normal(shape) * 0.05
        else:
            fan_in = int(np.prod(shape[1:]))
            v = rng.standard_normal(shape) * np.sqrt(2.0 / fan_in) * (0.25 if "branch2c" in name else 1.0)
        blobs[prefix + name] = v.astype(np.float32)
        trainable = name.startswith("fpn_") or (name.endswith("_w") and name[:4] in ("res3", "res4", "res5"))
        if momentum and trainable:
            blobs[prefix + name + "_momentum"] = (rng.standard_normal(shape) * 1e-3).astype(np.float32)
    return blobs, g


@pytest.mark.parametrize("arch", ["r50", "x101-64x4d"])
def test_backbone_blob_names_and_shapes_are_the_reference_builders(arch):
    """utils/net.backbone_blob_names against the parameter list of the imported reference builder
    (ResNet.py:85-130,221-283 + FPN.py:116-250 run under a recording model): same blobs, and the
    native network's filter shapes are the blobs' shapes."""
    import json
    from ssad_amd.backbone_pipeline import NativeResNetFPN
    g = json.load(open(os.path.join(GOLDEN, GRAPHS[arch])))
    shapes = {p["name"]: tuple(p["shape"]) for p in g["params"]}
    names = net.backbone_blob_names(arch)
    mine = [n for t in names.values() for n in t if n is not None]
    assert len(mine) == len(set(mine)) and set(mine) == set(shapes)
    nat = NativeResNetFPN.__new__(NativeResNetFPN)
    nat.arch, nat.train, nat.D = arch, arch != "x101-64x4d", 256
    nat._layers = OrderedDict()
    nat._define_layers()
    assert set(nat._layers) == set(names) and len(names) == len(nat._layers)
    for lname, (wn, sn, bn) in names.items():
        l = nat._layers[lname]
        assert shapes[wn] == (l.cout, l.wcin, l.k, l.k), (lname, wn)
        assert shapes[bn] == (l.cout,) and (sn is None or shapes[sn] == (l.cout,))
        assert (sn is not None) == l.affine
    # R-101 has no capture of its own: same rule, 23 blocks in res4
    r101 = net.backbone_blob_names("r101")
    assert r101["lat.1"][0] == "fpn_inner_res4_22_sum_lateral_w" and r101["out.1"][2] == "fpn_res4_22_sum_b"
    assert "res4.22.c3" in r101 and "res4.23.c1" not in r101 and r101["res4.0.proj"][0] == "res4_0_branch1_w"


def test_backbone_fold_and_unfold_round_trip_cpu():
    """Reference layout -> folded native parameters (W' = s W, m' = s m) -> reference layout: every
    blob comes back (filters and update history to fp32 rounding of the fold / un-fold, the
    AffineChannel blobs exactly), only the reference's trainable blobs carry `_momentum`, and the
    trainable folded filters' SGD row scales are s^2."""
    from ssad_amd.backbone_pipeline import NativeResNetFPN
    rng = np.random.default_rng(3)
    blobs, _ = reference_backbone_blobs("r50", rng)
    state, scales, moms, missing = net.backbone_from_blobs(blobs, "r50")
    assert not missing and len(scales) == 53                      # one AffineChannel per body convolution
    w, sc = blobs["res4_1_branch2b_w"], blobs["res4_1_branch2b_bn_s"]
    assert np.array_equal(state["res4.1.c2.weight"].numpy(), w * sc.reshape(-1, 1, 1, 1))
    assert np.array_equal(state["res4.1.c2.bias"].numpy(), blobs["res4_1_branch2b_bn_b"])
    assert np.array_equal(moms["res4.1.c2.weight"].numpy(),
                          blobs["res4_1_branch2b_w_momentum"] * sc.reshape(-1, 1, 1, 1))
    assert np.array_equal(state["lat.0.weight"].numpy(), blobs["fpn_inner_res5_2_sum_w"])   # no fold for the FPN
    with pytest.raises(KeyError):
        bad = dict(blobs)
        del bad["res2_0_branch1_bn_s"]
        net.backbone_from_blobs(bad, "r50")
    nat = NativeResNetFPN("r50", 1, (128, 128), "cpu", train=True, src=state, affine_scales=scales)
    net.load_backbone(nat, blobs)
    l = nat._layers["res4.1.c2"]
    assert l.s2 is not None and np.allclose(l.s2.numpy(), sc * sc, rtol=1e-6)
    assert nat._layers["res2.0.c1"].s2 is None                       # frozen: no update, no slot
    out = net.backbone_to_blobs(nat)
    assert set(out) == set(blobs)
    for k, v in blobs.items():
        if k.endswith("_bn_s") or k.endswith("_bn_b") or k.endswith("_b") or k.startswith("fpn_"):
            assert np.array_equal(out[k], v), k
        else:
            assert np.allclose(out[k], v, rtol=3e-7, atol=1e-12), k
    # another set of scales into the same network (a second checkpoint): slots exist for every trainable layer
    blobs2, _ = reference_backbone_blobs("r50", np.random.default_rng(4))
    net.load_backbone(nat, blobs2)
    assert np.allclose(l.s2.numpy(), blobs2["res4_1_branch2b_bn_s"] ** 2, rtol=1e-6)
    assert np.array_equal(net.backbone_to_blobs(nat)["res3_0_branch1_bn_s"], blobs2["res3_0_branch1_bn_s"])
    # a network built from the random initialisation has slots only where its own scale is not 1
    rnd = NativeResNetFPN("r50", 1, (128, 128), "cpu", train=True)
    from ssad_amd.kernels import KernelError
    with pytest.raises(KernelError):
        net.load_backbone(rnd, blobs)


def test_body_only_weights_file_loads_like_the_reference_cpu():
    """the reference's standard TRAIN.WEIGHTS is an ImageNet body (R-50.pkl: conv1 / res* blobs, no
    fpn_* / retnet_*).  initialize_gpu_from_weights_file logs '<name> not found' and keeps the initialised
    value (net.py:96-99); so does load_backbone(strict=False): the body is loaded, every FPN layer keeps its
    initialisation and its names come back as missing; a blob of another shape is skipped the same way;
    strict=True raises."""
    from ssad_amd.backbone_pipeline import NativeResNetFPN
    rng = np.random.default_rng(5)
    blobs, _ = reference_backbone_blobs("r50", rng)
    body = OrderedDict((k, v) for k, v in blobs.items() if not k.startswith("fpn_"))
    assert len(body) < len(blobs) and "conv1_w" in body
    state, scales, moms, missing = net.backbone_from_blobs(body, "r50", strict=False)
    assert "lat.0.weight" not in state and "fpn_6_w" in missing and "res5_2_branch2c_w" not in missing
    nat = NativeResNetFPN("r50", 1, (128, 128), "cpu", train=True, src=state, affine_scales=scales)
    fpn_before = {n: nat._layers[n].w.clone() for n in ("lat.0", "out.2", "p6", "p7")}
    assert all(torch.isfinite(v).all() and float(v.abs().max()) > 0 for v in fpn_before.values())
    net.load_backbone(nat, body)                                   # default: the reference's behaviour
    assert sorted(nat.missing_blobs) == sorted(k for k in blobs if k.startswith("fpn_") and not k.endswith("_momentum"))
    for n, v in fpn_before.items():
        assert torch.equal(nat._layers[n].w, v)                   # kept
    sc = body["res4_1_branch2b_bn_s"]
    assert np.array_equal(nat._layers["res4.1.c2"].w.numpy(), body["res4_1_branch2b_w"] * sc.reshape(-1, 1, 1, 1))
    assert np.allclose(nat._layers["res4.1.c2"].s2.numpy(), sc * sc, rtol=1e-6)
    with pytest.raises(KeyError):
        net.load_backbone(nat, body, strict=True)
    # a filter of another shape: logged and skipped (net.py:106-119), the layer keeps what it had
    odd = OrderedDict(body)
    odd["res3_0_branch2a_w"] = np.zeros((64, 256, 1, 1), np.float32)
    kept = nat._layers["res3.0.c1"].w.clone()
    net.load_backbone(nat, odd)
    assert "res3_0_branch2a_w" in nat.missing_blobs and torch.equal(nat._layers["res3.0.c1"].w, kept)
    with pytest.raises((ValueError, KeyError)):
        net.load_backbone(nat, odd, strict=True)


def _bcast_worker(rank, world, port, outdir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ssad_amd.backbone_pipeline import NativeResNetFPN
    blobs, _ = reference_backbone_blobs("r50", np.random.default_rng(6))
    state, scales, _, _ = net.backbone_from_blobs(blobs, "r50")
    ones = {k: torch.ones_like(v) for k, v in scales.items()}
    kw = dict(process_group=dist.group.WORLD, world_size=world)
    # rank 0 holds the file's weights; rank 1 was built from something else (slots present, other values)
    student = NativeResNetFPN("r50", 1, (128, 128), "cpu", train=True, src=state if rank == 0 else
                              {k: torch.zeros_like(v) for k, v in state.items()},
                              affine_scales=scales if rank == 0 else ones, **kw)
    teacher = NativeResNetFPN("r50", 1, (128, 128), "cpu", train=False, src=state if rank == 0 else None, **kw)
    if rank == 0:
        student.moms_flat.normal_(generator=torch.Generator().manual_seed(1))
    student.broadcast_params()
    teacher.broadcast_params()
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), params=student.params_flat.numpy(),
             moms=student.moms_flat.numpy(), frozen=student.frozen_flat.numpy(),
             s2=student._layers["res4.1.c2"].s2.numpy(), t_frozen=teacher.frozen_flat.numpy(),
             aff=student.affine_scale_values["res4.1.c2"].numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_covers_frozen_values_scale_slots_and_teacher_gloo():
    """broadcast_parameters covers all of model.params (net.py:185-208).  A replica that did not
    load the file must end with rank 0's frozen filters / folded biases, s^2 row scales and teacher."""
    import socket
    import tempfile
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_bcast_worker, args=(2, port, d), nprocs=2, join=True)
        r = [np.load(os.path.join(d, "rank%d.npz" % i)) for i in range(2)]
    for k in ("params", "moms", "frozen", "s2", "t_frozen", "aff"):
        assert np.array_equal(r[0][k], r[1][k]), k
    assert np.any(r[0]["frozen"] != 0) and np.any(r[0]["moms"] != 0) and np.any(r[0]["s2"] != 1)
