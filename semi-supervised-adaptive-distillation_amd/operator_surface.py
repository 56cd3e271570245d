"""The subnets' training iteration on the DROP-IN route: the reference's own graph, built by the
graph builders (modeling/retinanet_heads.py = detectron/lib/modeling/retinanet_heads.py:63-352),
differentiated by `AddGradientOperators` (caffe2/python/core.py), extended by the gradient exchange
and update operators of detectron/lib/modeling/optimizer.py:72-130, created once with
`workspace.CreateNet` and run with ONE `workspace.RunNet` per net and iteration -- what
detectron/tools/train_net.py:165-189 does.  Nothing here touches head_pipeline / ssad_program: every
launch is made by a registered Operator<HIPContext> (csrc/ops/) that the C-ABI created from the
serialized NetDef.

`bench.py` times this object as `also.operator_surface` beside the hand-built program
(head_pipeline.DistillHeads); tests compare both with the oracle.
"""
import numpy as np

from . import synth
from .caffe2_hip import caffe2_pb2, core, workspace
from .modeling import optimizer as opt
from .modeling import retinanet_heads as rh


class HeadsNetStep(object):
    """teacher net (test mode) + student training net over fed FPN features.

    feeds: every blob is a workspace blob on the GPU (FeedBlob copies host arrays once, outside any
    timed region); parameters, momenta, lr/one/wd live in the workspace and are updated in place by
    MomentumSGDUpdate, exactly as in the reference."""

    def __init__(self, cfg=None, N=2, shapes=synth.LEVEL_SHAPES_600, student_init=None, teacher_init=None,
                 lr=0.01, momentum=0.9, weight_decay=1e-4, gpu_id=0, update=True, allreduce=True,
                 lowering=True, prefix=""):
        self.cfg = cfg or rh.HeadConfig(num_gpus=1)
        self.N, self.shapes, self.update = N, list(shapes), update
        self.levels = list(self.cfg.levels())
        assert len(self.levels) == len(self.shapes)
        self.dev = core.DeviceOption(caffe2_pb2.HIP, gpu_id)
        cfg = self.cfg
        with core.DeviceScope(self.dev):
            self.teacher = rh.HeadModel(cfg, train=False, name=prefix + "teacher")
            rh.add_fpn_retinanet_outputs(self.teacher, ["teacher/fpn_%d" % l for l in reversed(self.levels)],
                                         cfg.fpn_dim, "teacher/")
            self.student = rh.HeadModel(cfg, train=True, name=prefix + "student")
            rh.add_fpn_retinanet_outputs(self.student, ["fpn_%d" % l for l in reversed(self.levels)], cfg.fpn_dim)
            loss_grads = rh.add_fpn_retinanet_losses(self.student)
            loss_grads.update(rh.add_distill_loss(self.student))
            self.forward_ops = len(self.student.net.Proto().op)
            self.grad_map = self.student.net.AddGradientOperators(loss_grads)
            self.backward_ops = len(self.student.net.Proto().op) - self.forward_ops
            self.init_blobs = {}
            if update:
                if allreduce:
                    opt.add_allreduce_ops(self.student, self.grad_map)
                self.init_blobs = opt.add_param_update_ops(self.student, self.grad_map, momentum, weight_decay)
                self.init_blobs["lr"] = np.full(1, lr, np.float32)
        for net in (self.teacher.net, self.student.net):
            net.Proto().type = "dag"                    # detector.py:66: cfg.MODEL.EXECUTION_TYPE
            net.Proto().num_workers = 4                 # detector.py:67
            if not lowering:
                net.Proto().arg.append(core.MakeArgument("hip_lowering", 0))
        self.total_ops = len(self.student.net.Proto().op) + len(self.teacher.net.Proto().op)
        rng = np.random.default_rng(1234)
        self.student_init = student_init if student_init is not None else synth.head_params(rng)
        self.teacher_init = teacher_init if teacher_init is not None else synth.head_params(rng)
        self._created = False

    # -- feeding ---------------------------------------------------------------------------------
    def feed_params(self):
        with core.DeviceScope(self.dev):
            for k, v in self.student_init.items():
                workspace.FeedBlob(k, v)
            for k, v in self.teacher_init.items():
                workspace.FeedBlob("teacher/" + k, v)
            for k, v in self.init_blobs.items():
                workspace.FeedBlob(k, v)

    def feed_inputs(self, fpn, teacher_fpn, labels, bbox_targets, fg_num):
        """fpn / teacher_fpn / labels: per level (finest first) numpy arrays; bbox_targets: per level
        (Y [M,4], L [M,4]); fg_num: scalar array."""
        with core.DeviceScope(self.dev):
            workspace.FeedBlob("retnet_fg_num", np.asarray(fg_num, np.float32).reshape(()))
            for i, l in enumerate(self.levels):
                workspace.FeedBlob("fpn_%d" % l, fpn[i])
                workspace.FeedBlob("teacher/fpn_%d" % l, teacher_fpn[i])
                workspace.FeedBlob("retnet_cls_labels_fpn%d" % l, labels[i])
                workspace.FeedBlob("retnet_roi_bbox_targets_fpn%d" % l, bbox_targets[i][0])
                workspace.FeedBlob("retnet_roi_fg_bbox_locs_fpn%d" % l, bbox_targets[i][1])

    def feed_synthetic(self, seed=1234):
        rng = np.random.default_rng(seed)
        fpn = synth.fpn_features(rng, self.N, self.shapes, self.cfg.fpn_dim)
        tfpn = synth.fpn_features(rng, self.N, self.shapes, self.cfg.fpn_dim)
        A, C = self.cfg.num_anchors, self.cfg.num_classes - 1
        labels = []
        for h, w in self.shapes:
            u = rng.random((self.N, A, h, w))
            lab = np.zeros((self.N, A, h, w), np.int32)
            lab[u < 0.05] = -1
            fg = (u >= 0.05) & (u < 0.07)
            lab[fg] = rng.integers(1, C + 1, size=int(fg.sum()), dtype=np.int32)
            labels.append(lab)
        tg = [synth.bbox_targets(rng, l) for l in labels]
        fg_num = np.array([max(1.0, float(sum(t[0].shape[0] for t in tg)))], np.float32)
        self.feed_inputs(fpn, tfpn, labels, tg, fg_num)
        return fpn, tfpn, labels, tg, fg_num

    # -- nets --------------------------------------------------------------------------------------
    def create(self):
        workspace.CreateNet(self.teacher.net, overwrite=True)
        workspace.CreateNet(self.student.net, overwrite=True)
        self._created = True
        return self

    def step(self, sync_every_op=False):
        """One iteration: two RunNet calls, each one synchronisation (train_net.py:173 runs one net
        because the reference appends the teacher's operators to the same net; here the frozen teacher
        is its own net, as model_builder.py:373-411 builds it)."""
        assert self._created, "create() first"
        workspace.RunNet(self.teacher.net, sync_every_op=sync_every_op)
        workspace.RunNet(self.student.net, sync_every_op=sync_every_op)

    # -- learning rate (detector.py:594-648) -------------------------------------------------------
    SCALE_MOMENTUM = True                 # config.py:634
    SCALE_MOMENTUM_THRESHOLD = 1.1        # config.py:638

    def update_lr(self, new_lr):
        """UpdateWorkspaceLr + _SetNewLr + _CorrectMomentum: the workspace's `lr` blob is the one source
        of truth; when it changes by more than the threshold every `<param>_momentum` is rescaled by
        new / old with one Scale operator per parameter (RunOperatorOnce), as the reference does."""
        cur_lr = np.float32(workspace.FetchBlob("lr")[0])
        new_lr = np.float32(new_lr)
        if cur_lr != new_lr:
            with core.DeviceScope(self.dev):
                workspace.FeedBlob("lr", np.array([new_lr], np.float32))
                eps = 1e-10
                ratio = max(new_lr / max(cur_lr, eps), cur_lr / max(new_lr, eps))
                if self.SCALE_MOMENTUM and cur_lr > 1e-7 and ratio > self.SCALE_MOMENTUM_THRESHOLD:
                    for p in opt.trainable_params(self.student):
                        workspace.RunOperatorOnce(core.CreateOperator(
                            "Scale", [p + "_momentum"], [p + "_momentum"], scale=float(new_lr / cur_lr)))
        return new_lr

    def lowered(self):
        return {"teacher": workspace.LoweredOps(self.teacher.net), "student": workspace.LoweredOps(self.student.net)}

    def losses(self):
        out = {}
        for l in self.levels:
            out["fl_distill_fpn%d" % l] = float(workspace.FetchBlob("fl_distill_fpn%d" % l))
            out["fl_fpn%d" % l] = float(workspace.FetchBlob("fl_fpn%d" % l))
            out["retnet_loss_bbox_fpn%d" % l] = float(workspace.FetchBlob("retnet_loss_bbox_fpn%d" % l))
        return out
