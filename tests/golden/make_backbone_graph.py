#!/usr/bin/env python3
"""Capture the operator list the REFERENCE's ResNet-50-FPN body builder emits
(SURVEY.md 8f row f1): detectron/lib/modeling/ResNet.py:47-260 +
FPN.py:60-260 for the RetinaNet configuration (FPN levels 3-7, extra conv
levels, stride on the first 1x1, frozen-BN AffineChannel, res2 frozen).

Runs ONLY in the build container (imports the reference's ResNet.py / FPN.py with
the stubs of make_head_graph.py).  The recording model restates what
CNNModelHelper.Conv / MaxPool (caffe2/python/helpers/{conv,pooling}.py with
use_cudnn=True, order=NCHW) and DetectionModelHelper.AffineChannel / ConvAffine
(detectron/lib/modeling/detector.py:83-107, 559-587) hand to the net.
Output: tests/golden/backbone_graph_{r50,x101_64x4d}_fpn.json (data only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_backbone_graph.py
"""
import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_head_graph as mh  # noqa: E402



class RecBackboneModel(mh.RecModel):
    def Conv(self, blob_in, blob_out, dim_in, dim_out, kernel, weight_init=None, bias_init=None,
             no_bias=0, group=1, **kwargs):
        kwargs = self._cudnn_kwargs(kwargs)
        if group != 1:                                                   # helpers/conv.py:124-125
            kwargs["group"] = group
        w = self.net._scoped(blob_out + "_w")
        winit = weight_init if weight_init else ("XavierFill", {})       # helpers/conv.py:83-93
        self.params.append({"name": w, "shape": [dim_out, int(dim_in / group), kernel, kernel],
                            "init": [winit[0], mh.plain(winit[1])]})
        ins = [blob_in, w]
        if not no_bias:
            b = self.net._scoped(blob_out + "_b")
            binit = bias_init if bias_init else ("ConstantFill", {})
            self.params.append({"name": b, "shape": [dim_out], "init": [binit[0], mh.plain(binit[1])]})
            ins.append(b)
        return self.net.Conv(ins, blob_out, kernel=kernel, order=self.order, **kwargs)

    def AffineChannel(self, blob_in, blob_out, dim, inplace=False):
        s, b = self.net._scoped(blob_out + "_s"), self.net._scoped(blob_out + "_b")
        self.params.append({"name": s, "shape": [dim], "init": ["ConstantFill", {"value": 1.0}]})
        self.params.append({"name": b, "shape": [dim], "init": ["ConstantFill", {"value": 0.0}]})
        return self.net.AffineChannel([blob_in, s, b], blob_in if inplace else blob_out)

    def ConvAffine(self, blob_in, prefix, dim_in, dim_out, kernel, stride, pad, group=1, dilation=1,
                   weight_init=None, bias_init=None, suffix="_bn", inplace=False):
        conv_blob = self.Conv(blob_in, prefix, dim_in, dim_out, kernel, stride=stride, pad=pad,
                              group=group, dilation=dilation, weight_init=weight_init,
                              bias_init=bias_init, no_bias=1)
        return self.AffineChannel(conv_blob, prefix + suffix, dim=dim_out, inplace=inplace)

    def MaxPool(self, blob_in, blob_out, **kwargs):
        # helpers/pooling.py:9-22 (use_cudnn -> engine CUDNN)
        kwargs["engine"] = "CUDNN"
        return self.net.MaxPool(blob_in, blob_out, order=self.order, **kwargs)

    def StopGradient(self, blob_in, blob_out):
        return self.net.StopGradient(blob_in, blob_out)


def install_backbone_stubs():
    """FPN.py also imports the reference's utils.c2 (caffe2.python), utils.boxes (compiled
    cython) and modeling.generate_anchors (np.float): the builder only uses the two filler
    helpers of utils.c2 (utils/c2.py:100-111), restated here."""
    import types
    import utils  # the reference package, for submodule registration
    c2 = types.ModuleType("utils.c2")
    c2.const_fill = lambda value: ("ConstantFill", {"value": value})
    c2.gauss_fill = lambda std: ("GaussianFill", {"std": std})
    sys.modules["utils.c2"] = c2
    utils.c2 = c2
    boxes = types.ModuleType("utils.boxes")
    sys.modules["utils.boxes"] = boxes
    utils.boxes = boxes
    ga = types.ModuleType("modeling.generate_anchors")
    ga.generate_anchors = None
    sys.modules["modeling.generate_anchors"] = ga


CASES = {
    # output file -> (yaml the settings come from, body builder, RESNETS overrides)
    "backbone_graph_r50_fpn.json": (
        "configs/focal_distillation/retinanet_R-50-FPN_distillation.yaml (body)",
        "add_fpn_ResNet50_conv5_body", {}),
    # the ResNeXt teacher (yaml :3,:19-24): stride on the 3x3, 64 groups of width 4
    "backbone_graph_x101_64x4d_fpn.json": (
        "configs/focal_distillation/retinanet_X-101-64x4d-FPN_1x_teacher.yaml (body)",
        "add_fpn_ResNet101_conv5_body",
        {"STRIDE_1X1": False, "NUM_GROUPS": 64, "WIDTH_PER_GROUP": 4}),
}


def main():
    mh.install_stubs()
    sys.path.insert(0, mh.REF)
    install_backbone_stubs()
    from core.config import cfg
    import modeling.FPN as FPN
    from collections import Counter

    cfg.FPN.FPN_ON = True
    cfg.FPN.MULTILEVEL_RPN = True
    cfg.FPN.RPN_MAX_LEVEL, cfg.FPN.RPN_MIN_LEVEL = 7, 3
    cfg.FPN.COARSEST_STRIDE = 128
    cfg.FPN.EXTRA_CONV_LEVELS = True
    cfg.RETINANET.RETINANET_ON = True
    # the py2 config stores names as bytes; under py3 the lookup key must be text
    cfg.RESNETS.TRANS_FUNC = "bottleneck_transformation"
    defaults = {k: getattr(cfg.RESNETS, k) for k in ("STRIDE_1X1", "NUM_GROUPS", "WIDTH_PER_GROUP")}

    for fname, (config, builder, overrides) in CASES.items():
        for k, v in defaults.items():
            setattr(cfg.RESNETS, k, overrides.get(k, v))
        model = RecBackboneModel(train=True)
        blobs, dim, scales = getattr(FPN, builder)(model)
        out = {
            "config": config,
            "resnets": {k: getattr(cfg.RESNETS, k) for k in defaults},
            "ops": model.ops,
            "params": model.params,
            "fpn_blobs": [str(b) for b in blobs],
            "fpn_dim": int(dim),
            "spatial_scales": [float(s) for s in scales],
            "op_histogram": dict(Counter(o["type"] for o in model.ops)),
        }
        with open(os.path.join(HERE, fname), "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)
        print(fname, out["op_histogram"], len(model.params), out["fpn_blobs"], out["spatial_scales"])


if __name__ == "__main__":
    main()
