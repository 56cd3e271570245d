#!/usr/bin/env python3
"""Capture the operator list the REFERENCE's graph builder emits for the
RetinaNet heads + distillation losses.

Runs ONLY in the build container: it imports
/root/reference/detectron/lib/modeling/retinanet_heads.py (py2 code that
imports under py3 once `past`, `cPickle`, `urllib2`, `cv2` and `caffe2.proto`
are stubbed) and drives its add_fpn_retinanet_outputs /
add_fpn_retinanet_losses / add_distill_loss with a recording model.  The
recording model restates what CNNModelHelper.Conv (caffe2/python/helpers/
conv.py:28-149 with use_cudnn=True, order=NCHW) and
DetectionModelHelper.ConvShared (detectron/lib/modeling/detector.py:449-482)
pass on to `net.Conv`.  Output: tests/golden/head_graph_r50_distill.json
(data only: op type / inputs / outputs / args, plus the parameter
initialisers).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_head_graph.py
"""
import json
import os
import pickle
import sys
import types

sys.dont_write_bytecode = True
REF = "/root/reference/detectron/lib"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "head_graph_r50_distill.json")


def install_stubs():
    past = types.ModuleType("past")
    builtins_ = types.ModuleType("past.builtins")
    builtins_.basestring = str
    past.builtins = builtins_
    sys.modules["past"], sys.modules["past.builtins"] = past, builtins_
    sys.modules["cPickle"] = pickle
    sys.modules["urllib2"] = types.ModuleType("urllib2")
    sys.modules["cv2"] = types.ModuleType("cv2")
    c2 = types.ModuleType("caffe2")
    proto = types.ModuleType("caffe2.proto")
    pb2 = types.ModuleType("caffe2.proto.caffe2_pb2")
    proto.caffe2_pb2 = pb2
    c2.proto = proto
    sys.modules["caffe2"], sys.modules["caffe2.proto"] = c2, proto
    sys.modules["caffe2.proto.caffe2_pb2"] = pb2


def plain(v):
    import numpy as np
    if isinstance(v, (np.floating,)):
        return float(v)
    if isinstance(v, (np.integer,)):
        return int(v)
    if isinstance(v, (list, tuple)):
        return [plain(x) for x in v]
    if isinstance(v, bytes):
        return v.decode()
    return v


class RecNet(object):
    """Records net.<Op>(...) calls.  `scope` restates caffe2's NameScope: plain
    string blob names are prefixed with the current name scope
    (caffe2/python/core.py ScopedBlobReference); the teacher is built under
    c2_utils.NamedTeacherScope() = "teacher/" (model_builder.py:384)."""

    def __init__(self, ops, scope=""):
        self.ops = ops
        self.scope = scope

    def _scoped(self, name):
        name = str(name)
        return name if (not self.scope or name.startswith(self.scope)) else self.scope + name

    def __getattr__(self, op_type):
        if op_type.startswith("__"):
            raise AttributeError(op_type)

        def add(inputs, outputs=None, **kw):
            ins = [self._scoped(i) for i in (inputs if isinstance(inputs, (list, tuple)) else [inputs])]
            outs = [self._scoped(o) for o in (outputs if isinstance(outputs, (list, tuple)) else [outputs])]
            self.ops.append({"type": op_type, "input": ins, "output": outs,
                             "args": {k: plain(v) for k, v in sorted(kw.items())}})
            return outs[0] if len(outs) == 1 else tuple(outs)
        return add


class RecModel(object):
    """What CNNModelHelper(use_cudnn=True, order='NCHW') + DetectionModelHelper
    forward to the net for the calls retinanet_heads.py makes."""

    def __init__(self, train, num_classes=81, scope=""):
        self.ops, self.params, self.losses, self.metrics = [], [], [], []
        self.net = RecNet(self.ops, scope)
        self.train = train
        self.num_classes = num_classes
        self.order = "NCHW"
        self.use_cudnn = True
        self.cudnn_exhaustive_search = False
        self.ws_nbytes_limit = None

    def _cudnn_kwargs(self, kwargs):
        kwargs["engine"] = "CUDNN"
        kwargs["exhaustive_search"] = self.cudnn_exhaustive_search
        if self.ws_nbytes_limit:
            kwargs["ws_nbytes_limit"] = self.ws_nbytes_limit
        return kwargs

    def Conv(self, blob_in, blob_out, dim_in, dim_out, kernel, weight_init=None,
             bias_init=None, **kwargs):
        kwargs = self._cudnn_kwargs(kwargs)
        w, b = self.net._scoped(blob_out + "_w"), self.net._scoped(blob_out + "_b")
        self.params.append({"name": w, "shape": [dim_out, dim_in, kernel, kernel],
                            "init": [weight_init[0], plain(weight_init[1])]})
        self.params.append({"name": b, "shape": [dim_out],
                            "init": [bias_init[0], plain(bias_init[1])]})
        return self.net.Conv([blob_in, w, b], blob_out, kernel=kernel, order=self.order, **kwargs)

    def ConvShared(self, blob_in, blob_out, dim_in, dim_out, kernel, weight=None, bias=None,
                   **kwargs):
        kwargs = self._cudnn_kwargs(kwargs)
        return self.net.Conv([blob_in, weight, bias], blob_out, kernel=kernel, order=self.order,
                             **kwargs)

    def Relu(self, blob_in, blob_out):
        return self.net.Relu(blob_in, blob_out)

    def GetLossScale(self):
        from core.config import cfg
        return 1.0 / cfg.NUM_GPUS

    def AddLosses(self, losses):
        self.losses.extend([str(l) for l in (losses if isinstance(losses, list) else [losses])])

    def AddMetrics(self, metrics):
        self.metrics.extend([str(m) for m in (metrics if isinstance(metrics, list) else [metrics])])


def main():
    install_stubs()
    sys.path.insert(0, REF)
    from core.config import cfg
    import modeling.retinanet_heads as rh

    # values of configs/focal_distillation/retinanet_R-50-FPN_distillation.yaml
    cfg.NUM_GPUS = 8
    cfg.MODEL.NUM_CLASSES = 81
    cfg.FPN.RPN_MAX_LEVEL, cfg.FPN.RPN_MIN_LEVEL = 7, 3
    cfg.RETINANET.NUM_CONVS = 4
    cfg.RETINANET.ASPECT_RATIOS = (1.0, 2.0, 0.5)
    cfg.RETINANET.SCALES_PER_OCTAVE = 3
    cfg.RETINANET.LOSS_GAMMA, cfg.RETINANET.LOSS_ALPHA = 2.0, 0.25
    cfg.DISTILLATION.LOSS_ALPHA, cfg.DISTILLATION.LOSS_GAMMA = 0.5, 2.0
    cfg.DISTILLATION.IGNORED_LABEL = -1
    cfg.DISTILLATION.ADAPTIVE_NORMALIZER = True
    cfg.DISTILLATION.LOGITS_POWER = 1.8
    cfg.DISTILLATION.TEMPERATURE = 1.0

    blobs_in = ["fpn_%d" % l for l in range(7, 2, -1)]   # coarsest first (FPN.py order)

    student = RecModel(train=True)
    rh.add_fpn_retinanet_outputs(student, blobs_in, 256, None)
    n_head = len(student.ops)
    loss_grads = {}
    loss_grads.update(rh.add_fpn_retinanet_losses(student))
    loss_grads.update(rh.add_distill_loss(student, "", "teacher/"))

    teacher = RecModel(train=False, scope="teacher/")
    rh.add_fpn_retinanet_outputs(teacher, ["teacher/" + b for b in blobs_in], 256, None)

    from collections import Counter
    out = {
        "config": "configs/focal_distillation/retinanet_R-50-FPN_distillation.yaml",
        "student_ops": student.ops,
        "student_head_op_count": n_head,
        "student_params": student.params,
        "student_losses": student.losses,
        "student_metrics": student.metrics,
        "loss_gradients": {str(k): str(v) for k, v in sorted(loss_grads.items())},
        "teacher_ops": teacher.ops,
        "op_histogram": dict(Counter(o["type"] for o in student.ops)),
        "teacher_op_histogram": dict(Counter(o["type"] for o in teacher.ops)),
    }
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(out["op_histogram"], out["teacher_op_histogram"], len(student.params))


if __name__ == "__main__":
    main()
